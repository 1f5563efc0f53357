"""CPU checks of the round-2 additions to the oracle (test infrastructure): IR-range neighbour sensing, the counter-based
reset spawn, the damping models, and the host logic around them (status surfacing, refusal of user-programmed kilobots)."""
import math
import warnings

import numpy as np
import pytest

from oracle import oracle as O
from tests import scenes


# ---------------------------------------------------------------------------------------------- random numbers
def test_philox_known_answers():
    """Random123's published known-answer vectors for philox4x32-10."""
    assert O.philox4x32_10((0, 0, 0, 0), (0, 0)) == (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)
    assert O.philox4x32_10((0xffffffff,) * 4, (0xffffffff,) * 2) == (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)
    assert O.philox4x32_10((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0)) == \
        (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)


def test_logf_against_libm():
    rng = np.random.RandomState(0)
    xs = np.concatenate([2.0 ** -24 * (1 + rng.randint(0, 2 ** 24, size=2000)), [2.0 ** -24, 0.5, 0.70710678, 1.0]])
    for x in xs.astype(np.float32):
        ref = math.log(float(x))
        assert abs(O.logf(float(x)) - ref) <= 2e-7 * max(1.0, abs(ref)), x


# ---------------------------------------------------------------------------------------------- reset
def test_reset_spawn_rule():
    """yaml_kilobots_env.py:346-352: N(mean, std), clipped to bounds -/+ 0.02, theta = 0; deterministic in the seed."""
    E, N = 64, 256
    o = O.OracleSim(O.default_config(E, N))
    o.reset(seed=7, mean=(0.1, -0.05), std=0.1, resolve=False)
    x, y = o.x.astype(np.float64) / 25.0, o.y.astype(np.float64) / 25.0
    assert abs(x.mean() - 0.1) < 3e-3 and abs(y.mean() + 0.05) < 3e-3
    assert abs(x.std() - 0.1) < 3e-3 and abs(y.std() - 0.1) < 3e-3
    assert abs(np.corrcoef(x.ravel(), y.ravel())[0, 1]) < 0.03
    assert np.all(o.theta == 0.0) and np.all(o.v == 0.0) and np.all(o.w == 0.0) and np.all(o.ws_cnt == 0)
    # a standard normal: fourth moment 3, tails present
    z = (x - 0.1) / 0.1
    assert abs((z ** 4).mean() - 3.0) < 0.15 and (np.abs(z) > 3).mean() > 1e-3
    x1 = o.x.copy()
    o.reset(seed=7, mean=(0.1, -0.05), std=0.1, resolve=False)
    assert np.array_equal(x1, o.x)
    o.reset(seed=8, mean=(0.1, -0.05), std=0.1, resolve=False)
    assert not np.array_equal(x1, o.x)
    # clipping: a wide cloud ends on the bounds -/+ 0.02 exactly
    o.reset(seed=1, std=2.0, resolve=False)
    assert o.x.max() == np.float32(np.float32(0.5 * 2.0 - 0.02) * np.float32(25.0))
    assert o.y.min() == np.float32(np.float32(-0.5 * 1.5 + 0.02) * np.float32(25.0))


def test_reset_random_theta_and_velocity():
    o = O.OracleSim(O.default_config(8, 512))
    o.reset(seed=3, random_theta=True, random_velocity=True, resolve=False)
    assert -math.pi <= o.theta.min() < -3.0 and 3.0 < o.theta.max() <= math.pi
    assert 0.0 <= o.v.min() and o.v.max() < 0.01 and o.v.mean() == pytest.approx(0.005, abs=3e-4)
    assert -math.pi / 2 <= o.w.min() and o.w.max() < math.pi / 2 and abs(o.w.mean()) < 0.05


def test_reset_shard_equals_rows_of_the_whole_batch():
    E, N = 7, 40
    whole = O.OracleSim(O.default_config(E, N))
    whole.reset(seed=11, std=0.2, random_theta=True)
    lo = 3
    part = O.OracleSim(O.default_config(E - lo, N))
    part.reset(seed=11, std=0.2, random_theta=True, env_offset=lo)
    for f in ('x', 'y', 'theta'):
        assert np.array_equal(getattr(whole, f)[lo:], getattr(part, f))


# ---------------------------------------------------------------------------------------------- sensing
def _brute(x, y, R):
    Rw = np.float32(np.float32(R) * np.float32(25.0))
    R2 = np.float32(Rw * Rw)
    dx = (x[:, None, :] - x[:, :, None]).astype(np.float32)
    dy = (y[:, None, :] - y[:, :, None]).astype(np.float32)
    dd = (dx * dx).astype(np.float32) + (dy * dy).astype(np.float32)
    within = ~(dd > R2)
    return within.sum(-1) - 1


@pytest.mark.parametrize('R', [0.034, 0.07, 0.1, 0.3])
def test_sense_counts(R):
    E, N = 3, 200
    xy, th = scenes.gaussian_spawn(E, N, sigma=0.25, seed=2)
    o = O.OracleSim(O.default_config(E, N))
    o.set_poses_m(xy, th)
    got = o.sense(R)
    assert np.array_equal(got, _brute(o.x, o.y, R))
    assert got.sum() % 2 == 0          # the relation is symmetric: every pair is counted at both ends


def test_step_writes_the_counts_of_the_sensing_point():
    """kb_config.sense_radius: nbr_count holds what the kilobots sensed at the START of the last substep, like the light."""
    E, N, R = 2, 64, 0.08
    xy, th = scenes.gaussian_spawn(E, N, sigma=0.12, seed=5)
    o = O.OracleSim(O.default_config(E, N, sense_radius=R))
    o.set_poses_m(xy, th)
    o.set_actions(scenes.random_actions(E, N, seed=1))
    o.step(3)
    before_last = o.sense(R).copy()
    o.step(1)
    assert np.array_equal(o.nbr_count, before_last)
    o.step(1, flags=O.STEP_NO_DRIVE)            # reset's world step does not sense (no drive phase)
    assert np.array_equal(o.nbr_count, before_last)


# ---------------------------------------------------------------------------------------------- damping model
def test_damping_models():
    """b2Island::Solve: Pade v / (1 + h c) (Box2D >= 2.3.1) vs clamp(1 - h c, 0, 1) (Box2D <= 2.3.0); contact-free substep."""
    res = {}
    for model in (0, 1):
        o = O.OracleSim(O.default_config(1, 1, damping_model=model))
        o.set_poses_m(np.zeros((1, 1, 2)), np.zeros((1, 1)))
        o.set_actions(np.array([[[0.008, 0.3]]], np.float32))
        o.step(1)
        res[model] = (float(o.x[0, 0]) / 25.0, float(o.theta[0, 0]))
    assert res[0][0] == pytest.approx(0.1 * 0.008 / 1.08, rel=2e-6) and res[0][1] == pytest.approx(0.1 * 0.3 / 1.08, rel=2e-6)
    assert res[1][0] == pytest.approx(0.1 * 0.008 * 0.92, rel=2e-6) and res[1][1] == pytest.approx(0.1 * 0.3 * 0.92, rel=2e-6)
    # a damping of 15 / s: the linear model clamps at zero, Pade never does
    o = O.OracleSim(O.default_config(1, 1, damping_model=1, bot_linear_damping=15.0))
    o.set_poses_m(np.zeros((1, 1, 2)), np.zeros((1, 1)))
    o.set_actions(np.array([[[0.008, 0.0]]], np.float32))
    o.step(1)
    assert o.x[0, 0] == 0.0


# ---------------------------------------------------------------------------------------------- host logic
def _env(std, n=60, **kw):
    import torch  # noqa: F401
    from tests.oracle_backend import OracleBackend
    from gym_kilobots_amd.envs import BatchedKilobotsEnv
    return BatchedKilobotsEnv(2, n, spawn_std=std, sim_factory=OracleBackend, **kw)


def test_capacity_overflow_is_loud():
    from gym_kilobots_amd import _native as nat
    # 60 kilobots on one spot with a contact store of 64 entries: contacts are dropped -> reset() must say so
    with pytest.raises(nat.KilobotsStatusError, match='contact capacity overflow'):
        _env(0.005, contact_capacity=64).reset()
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        _env(0.005, contact_capacity=64, on_status='warn').reset()
    assert any('contact capacity overflow' in str(x.message) for x in w)
    _env(0.005, contact_capacity=64, on_status='ignore').reset()
    # the default sizes the store from the spawn: the same pile is fine
    env = _env(0.005)
    env.reset()
    assert env.sim.status_bits() == 0
    assert env.sim.cfg.contact_capacity >= 60 * 59 // 2


def test_batched_reset_is_seeded_and_on_device_rule():
    a, b = _env(0.1, seed=4), _env(0.1, seed=4)
    assert np.array_equal(a.reset().numpy(), b.reset().numpy())
    assert not np.array_equal(a.reset().numpy(), b.sim.poses().numpy())      # the second reset draws new positions
    c = _env(0.1, seed=5)
    assert not np.array_equal(c.reset().numpy(), b.sim.poses().numpy())


def test_user_programmed_kilobots_run_their_loop_on_the_host():
    """a user _loop used to be ignored, then refused; now it runs (tests/test_host_programmed.py has the
    parity tests).  A subclass of a device-law class that overrides _loop replaces that law, as in the reference."""
    from gym_kilobots_amd.envs import KilobotsEnv
    from gym_kilobots_amd.lib import PhototaxisKilobot, CircularGradientLight, Kilobot
    from tests.oracle_backend import OracleBackend

    class MyBot(PhototaxisKilobot):
        def _loop(self):                       # reference extension point (kilobot.py:164-168): runs per kilobot per substep
            self.set_motors(10, 20)

    class Plain(Kilobot):                      # only _setup is user code: that runs on the host at construction
        def _setup(self):
            self.set_motors(0, 200)

        def _loop(self):
            pass

    def make(cls):
        class Env(KilobotsEnv):
            def _configure_environment(self):
                self._light = CircularGradientLight(position=np.zeros(2))
                self._add_kilobot(cls(self.world, position=(0.1, 0.0), light=self._light))

            def get_reward(self, *a):
                return 0.0
        return Env(sim_factory=OracleBackend)
    env = make(MyBot)
    env.reset()
    assert env._host_programmed and env.kilobots[0].get_motors() == (255, 0)        # _setup -> turn_left
    env.step(None)
    assert env.kilobots[0].get_motors() == (10, 20)
    env = make(Plain)
    env.reset()
    assert not env._host_programmed and env.kilobots[0].get_motors() == (0, 200)


def test_multi_env_light_state_has_the_single_env_layout():
    """with num_envs > 1 the state of a MomentumLight must keep its velocity, (x, y, vx, vy) per component,
    exactly like the single-env path (reference light.py:318-319, 256-257)."""
    from gym_kilobots_amd.envs import KilobotsEnv
    from gym_kilobots_amd.lib import SimplePhototaxisKilobot, MomentumLight, CircularGradientLight, CompositeLight
    from tests.oracle_backend import OracleBackend

    def make(num_envs, composite):
        class Env(KilobotsEnv):
            def _configure_environment(self):
                m = MomentumLight(position=np.array([0.1, -0.05]), radius=0.3)
                self._light = CompositeLight([CircularGradientLight(position=np.array([-0.2, 0.0]), radius=0.25), m]) if composite else m
                for x in (-0.1, 0.0, 0.1):
                    self._add_kilobot(SimplePhototaxisKilobot(self.world, position=(x, 0.2), light=self._light))

            def get_reward(self, *a):
                return 0.0
        return Env(num_envs=num_envs, sim_factory=OracleBackend)
    for composite in (False, True):
        one, many = make(1, composite), make(3, composite)
        one.reset()
        many.reset()
        adim = one.action_space.shape[0]
        a = np.linspace(0.004, 0.01, adim)
        for _ in range(3):
            s1, *_ = one.step(a)
            s3, *_ = many.step(a)
        assert s1['light'].shape == ((6,) if composite else (4,))
        assert s3['light'].shape == (3, 6 if composite else 4)
        assert np.allclose(s3['light'], s1['light'][None], atol=1e-7)
        assert np.abs(s1['light'][-2:]).max() > 0          # the momentum light moves
