"""Sleeping: b2World(gravity, doSleep=True) of the reference (kilobots_env.py:45), restated from Box2D 2.3.1
(b2Island::Solve allowSleep block, b2World::Solve island seeds, b2Body::SetAwake / SetLinearVelocity / SetAngularVelocity).

CPU: known-answer behaviour of the oracle.  GPU (-m gpu): the HIP path against the oracle, bit for bit, including the sleep
times, on every solver path, with objects, in fused launches."""
import numpy as np
import pytest
import torch

from oracle import oracle as O
from tests import scenes


def _sim(E, N, xy, th=None, **kw):
    s = O.OracleSim(O.default_config(E, N, O.DRIVE_VELOCITY, O.LIGHT_NONE, **kw))
    s.set_poses_m(np.asarray(xy, np.float64).reshape(E, N, 2), np.zeros((E, N)) if th is None else th)
    return s


def test_a_resting_body_sleeps_after_half_a_second_and_a_driven_one_never():
    s = _sim(1, 2, [[0.0, 0.0], [0.5, 0.3]], allow_sleep=1)
    a = np.zeros((1, 2, 2), np.float32)
    a[0, 1] = (0.01, 0.0)                      # kilobot 1 drives, kilobot 0 is commanded (0, 0)
    s.set_actions(a)
    times = []
    for _ in range(7):
        s.step(1)
        times.append(s.sleep_time[0].copy())
    t = np.array(times)
    # b2_timeToSleep = 0.5 s at dt = 0.1: m_sleepTime 0.1 .. 0.4, the fifth slow step puts the (one-body) island to sleep
    np.testing.assert_allclose(t[:4, 0], [0.1, 0.2, 0.3, 0.4], rtol=1e-6)
    assert (t[4:, 0] == -1.0).all()
    assert (t[:, 1] == 0.0).all()              # faster than b2_linearSleepTolerance: m_sleepTime stays 0
    # a non-zero command wakes it (b2Body::SetLinearVelocity), and it moves again
    x0 = s.x[0, 0]
    a[0, 0] = (0.01, 0.0)
    s.set_actions(a)
    s.step(1)
    assert s.sleep_time[0, 0] == 0.0 and s.x[0, 0] > x0
    # a command below the tolerance keeps it awake but lets the sleep time grow again: 0.0003 m/s * 25 * 0.926 < 0.01 units/s
    a[0, 0] = (0.0003, 0.0)
    s.set_actions(a)
    s.step(2)
    np.testing.assert_allclose(s.sleep_time[0, 0], 0.2, rtol=1e-6)


def test_sleeping_stops_the_residual_creep_of_a_resolved_overlap():
    def run(sleep):
        s = _sim(1, 3, [[0.0, 0.0], [0.02, 0.0], [0.5, 0.5]], allow_sleep=sleep)
        out = []
        for _ in range(14):
            s.step(1)
            out.append(s.x[0].copy())
        return np.array(out), s
    on, s_on = run(1)
    off, _ = run(0)
    assert np.array_equal(on[:5], off[:5])                          # identical until the island falls asleep
    assert (on[5:] == on[4]).all()                                   # asleep: the bodies do not move any more
    assert (np.abs(np.diff(off[4:, 0])) > 0).all()                   # without sleeping the position solver keeps creeping
    assert (s_on.sleep_time[0] == -1.0).all()
    # the contact impulses of the sleeping pair are carried over untouched
    assert s_on.ws_cnt[0].sum() == 1


def test_an_awake_body_wakes_the_island_it_touches():
    # kilobot 0 sleeps next to kilobot 1 (touching, at rest); kilobot 2 drives into kilobot 1 from the far side
    r = 0.0165
    s = _sim(1, 3, [[0.0, 0.0], [2 * r - 0.0005, 0.0], [4 * r + 0.02, 0.0]], th=np.array([[0.0, 0.0, np.pi]]), allow_sleep=1)
    a = np.zeros((1, 3, 2), np.float32)
    s.set_actions(a)
    s.step(8)
    assert (s.sleep_time[0] == -1.0).all()
    x_before = s.x[0].copy()
    a[0, 2] = (0.01, 0.0)
    s.set_actions(a)
    woke = None
    for k in range(40):
        s.step(1)
        if s.sleep_time[0, 0] >= 0.0:
            woke = k
            break
        assert s.x[0, 0] == x_before[0] and s.x[0, 1] == x_before[1]    # still asleep: not simulated
    assert woke is not None and woke > 3                                 # woken when the driver arrives, not before
    s.step(5)
    assert s.x[0, 0] < x_before[0]                                        # and pushed along


def test_an_island_that_is_not_solved_stays_awake():
    # a deep pile needs more than one step of position correction: awake while min separation < -3 slop
    xy = np.array([[0.001 * i, 0.0005 * (i % 3)] for i in range(12)])
    s = _sim(1, 12, xy, allow_sleep=1, pos_iters=1)
    s.step(5)
    assert (s.sleep_time[0] >= 0.0).all() and s.sleep_time[0].max() > 0.3
    s2 = _sim(1, 12, xy, allow_sleep=1, pos_iters=0)                      # no position iterations: positionSolved stays false
    s2.step(12)
    assert (s2.sleep_time[0] >= 0.0).all()


def test_sleeping_cannot_change_a_swarm_in_which_every_kilobot_is_commanded_to_move():
    """The benchmark workload (cfg3): fresh non-zero velocity commands for every kilobot in every substep.  A kilobot that
    fell asleep is woken by its next command before anything is simulated, and a kilobot's velocity does not outlive a
    substep, so the trajectories with and without the sleep state are the same bits (why the fixed-size benchmark
    instantiations may leave the bookkeeping out)."""
    E, N = 2, 256
    xy, th = scenes.lattice_spawn(E, N, seed=3, pitch=0.036)
    sims = [_sim(E, N, xy, th, allow_sleep=k) for k in (0, 1)]
    slept = False
    for k in range(30):
        a = scenes.random_actions(E, N, seed=50 + k)
        a[..., 0] = np.maximum(a[..., 0], 1e-4)
        for s in sims:
            s.set_actions(a)
            s.step(1)
        slept = slept or bool((sims[1].sleep_time < 0).any())
        for f in ('x', 'y', 'theta', 'ws_acc'):
            assert np.array_equal(getattr(sims[0], f), getattr(sims[1], f)), (k, f)
    assert sims[1].sleep_time.max() > 0.0          # blocked kilobots do accumulate sleep time ...


def test_a_pushed_disc_comes_to_rest_and_sleeps_with_zero_velocity():
    s = O.OracleSim(O.default_config(1, 2, O.DRIVE_VELOCITY, O.LIGHT_NONE, allow_sleep=1, num_objects=1))
    s.set_poses_m(np.array([[[-0.1, 0.0], [0.6, 0.4]]]), np.zeros((1, 2)))
    s.set_objects_m(np.array([[[0.0, 0.0]]]))
    a = np.zeros((1, 2, 2), np.float32)
    a[0, 0] = (0.01, 0.0)
    s.set_actions(a)
    s.step(120)                                    # drives into the disc and pushes it
    assert s.ox[0, 0] > 0.05 and s.osleep[0, 0] == 0.0           # (world units)
    a[0, 0] = (0.0, 0.0)
    s.set_actions(a)
    s.step(60)                                     # damping 0.8 / s: below the tolerance after a few seconds, asleep 0.5 s later
    assert s.osleep[0, 0] == -1.0 and s.ovx[0, 0] == 0.0 and s.ovy[0, 0] == 0.0 and s.ow[0, 0] == 0.0
    assert s.sleep_time[0, 0] == -1.0
    x = s.ox[0, 0]
    s.step(10)
    assert s.ox[0, 0] == x


# ---------------------------------------------------------------------------------------------------------------- GPU
def _pair(E, N, xy, th, objects=None, **kw):
    from tests.test_parity_gpu import make_pair
    return make_pair(E, N, O.DRIVE_VELOCITY, O.LIGHT_NONE, xy=xy, th=th, objects=objects, allow_sleep=1, **kw)


def _stop_and_go_actions(E, N, k, seed):
    """Velocity commands in which groups of kilobots stop for a while (they fall asleep in their islands) and start again."""
    rng = np.random.RandomState(seed + k)
    a = scenes.random_actions(E, N, seed=seed + 100 + k)
    phase = (np.arange(N)[None, :] // 7 + np.arange(E)[:, None] + k // 9) % 3
    a[phase == 0] = 0.0                                   # a third of the kilobots rests for 9 substeps at a time
    a[rng.rand(E, N) < 0.05] = 0.0
    return a.astype(np.float32)


SLEEP_FIELDS = ('x', 'y', 'theta', 'sleep_time')


@pytest.mark.gpu
@pytest.mark.parametrize('N,solver_mode', [(40, 0), (64, 0), (100, 0), (256, 0), (256, 1), (256, 2), (200, 3), (128, 4), (1024, 0)])
def test_sleeping_swarm_equals_oracle(N, solver_mode):
    from tests.test_parity_gpu import assert_same, assert_ws_same, dev
    E = 3
    xy, th = scenes.gaussian_spawn(E, N, sigma=0.05 + 0.0004 * N, seed=N + solver_mode)
    osim, gsim = _pair(E, N, xy, th, solver_mode=solver_mode)
    slept = woke = False
    prev = None
    for k in range(45):
        a = _stop_and_go_actions(E, N, k, seed=7)
        osim.set_actions(a)
        gsim.set_actions(dev(a))
        osim.step(1)
        gsim.step(1)
        assert_same(osim, gsim, 'N %d mode %d substep %d' % (N, solver_mode, k), SLEEP_FIELDS)
        asleep = osim.sleep_time < 0
        slept = slept or bool(asleep.any())
        if prev is not None:
            woke = woke or bool((prev & ~asleep).any())
        prev = asleep
    assert_ws_same(osim, gsim, 'N %d' % N)
    assert slept and woke
    assert int(gsim.status.max().item()) == 0 or N >= 256      # (dense Gaussian spawns may flag slot overflows; still compared)


@pytest.mark.gpu
def test_fused_launch_with_sleeping_equals_single_substeps_and_oracle():
    from tests.test_parity_gpu import assert_same, dev, make_pair
    E, N = 4, 96
    xy, th = scenes.gaussian_spawn(E, N, sigma=0.08, seed=11)
    osim, gsim = _pair(E, N, xy, th)
    _, gsim1 = make_pair(E, N, O.DRIVE_VELOCITY, O.LIGHT_NONE, xy=xy, th=th, allow_sleep=1)
    for k in range(6):
        a = _stop_and_go_actions(E, N, 9 * k, seed=3)
        osim.set_actions(a)
        osim.step(10)
        gsim.step(10, actions=dev(a))
        gsim1.set_actions(dev(a))
        for _ in range(10):
            gsim1.step(1)
        assert_same(osim, gsim, 'fused %d' % k, SLEEP_FIELDS)
        assert_same(osim, gsim1, 'single %d' % k, SLEEP_FIELDS)
    assert (osim.sleep_time < 0).any()


@pytest.mark.gpu
@pytest.mark.parametrize('boxes,N', [(False, 48), (True, 48), (True, 200), (False, 1024)])
def test_sleeping_with_objects_equals_oracle(boxes, N):
    from tests.test_parity_gpu import assert_same, dev, OBJ_FIELDS
    E = 2
    xy, th = scenes.gaussian_spawn(E, N, sigma=0.12 + 0.0002 * N, seed=5 + N)
    objs = np.tile(np.array([[0.1, 0.05], [-0.15, -0.1], [0.3, -0.25]])[None], (E, 1, 1))
    kw = {}
    if boxes:
        kw = dict(obj_shape=[1, 1, 0, 0, 0, 0, 0, 0], obj_nverts=[4, 4, 0, 0, 0, 0, 0, 0],
                  obj_verts=[[[0.075 * 25.0, 0.05 * 25.0]] + [[0.0, 0.0]] * 3] * 2 + [[[0.0, 0.0]] * 4] * 6)
    osim, gsim = _pair(E, N, xy, th, objects=objs, **kw)
    obj_slept = False
    for k in range(60):
        a = _stop_and_go_actions(E, N, k, seed=21)
        if k >= 30:
            a[...] = 0.0                                   # everything comes to rest: the objects fall asleep too
        osim.set_actions(a)
        gsim.set_actions(dev(a))
        osim.step(1)
        gsim.step(1)
        assert_same(osim, gsim, 'objects substep %d' % k, OBJ_FIELDS + ('sleep_time', 'osleep'))
        obj_slept = obj_slept or bool((osim.osleep < 0).any())
    # (islands whose position constraints have not converged stay awake however long they rest: b2Island::Solve positionSolved)
    assert (osim.sleep_time < 0).any()
    if N <= 200:
        assert obj_slept and (osim.sleep_time < 0).mean() > 0.5


@pytest.mark.gpu
def test_env_api_sleeps_like_the_reference_world():
    """KilobotsEnv creates its world like kilobots_env.py:45 (doSleep=True): the HIP-backed env and the oracle-backed env agree
    bit for bit over steps in which the swarm rests, and resting kilobots stop creeping."""
    from gym_kilobots_amd.envs import DirectControlKilobotsEnv
    from gym_kilobots_amd.lib import SimpleVelocityControlKilobot
    from tests.oracle_backend import OracleBackend

    class Crowd(DirectControlKilobotsEnv):
        def _configure_environment(self):
            rng = np.random.RandomState(2)
            for p in rng.normal(scale=0.04, size=(30, 2)):
                self._add_kilobot(SimpleVelocityControlKilobot(self.world, position=p, orientation=rng.uniform(-3, 3), velocity=[0.0, 0.0]))

        def get_reward(self, *a):
            return 0.

    g, o = Crowd(), Crowd(sim_factory=OracleBackend)
    g.reset(), o.reset()
    assert g.sim.cfg.allow_sleep == 1 and g.sim.sleep_time is not None
    rng = np.random.RandomState(4)
    last = None
    for k in range(8):
        a = None if k in (2, 3, 4, 6) else rng.uniform([0, -1.5], [0.01, 1.5], size=(30, 2))
        og, *_ = g.step(a)
        oo, *_ = o.step(a)
        assert np.array_equal(og['kilobots'], oo['kilobots']), k
        if k in (3, 4):
            assert np.array_equal(og['kilobots'], last)           # asleep since the first resting step: nothing creeps
        last = og['kilobots']
    assert np.array_equal(g.sim.sleep_time.cpu().numpy(), o.sim.sleep_time.numpy())
    g.close()


@pytest.mark.gpu
def test_benchmark_kernel_without_the_sleep_state_equals_the_sleeping_kernel_on_a_commanded_swarm():
    """cfg3 slice (1024 kilobots, every kilobot commanded to move in every substep): the fixed-size benchmark instantiation
    (no sleep state) and the instantiation with the sleep state give the same bits -- kilobots do fall asleep
    (pressed against their neighbours they rest for 0.5 s) and are woken by their next command before anything is simulated."""
    from tests.test_parity_gpu import dev, make_pair
    from gym_kilobots_amd.sim import KilobotSim
    E, N = 6, 1024
    xy, th = scenes.lattice_spawn(E, N, seed=9)
    plain = KilobotSim(E, N, allow_sleep=0)
    sleepy = KilobotSim(E, N, allow_sleep=1)
    for s in (plain, sleepy):
        s.set_poses_m(xy, th)
    slept = 0
    for k in range(60):
        a = scenes.random_actions(E, N, seed=300 + k)
        a[..., 0] = np.maximum(a[..., 0], 1e-4)
        for s in (plain, sleepy):
            s.step(1, actions=dev(a))
        slept += int((sleepy.sleep_time < 0).sum().item())
        for f in ('x', 'y', 'theta', 'ws_acc'):
            assert torch.equal(getattr(plain, f), getattr(sleepy, f)), (k, f)
    assert sleepy.sleep_time.max().item() > 0.0
    # (both are fixed-size 1024-kilobot instantiations, three envs per CU: with and without the sleep state)
    assert plain.lds_bytes == 52496 and sleepy.lds_bytes == 52496 and sleepy.resident_envs_per_cu == 3


def _creeper_scene():
    """Kilobot 0 creeps away from kilobot 1 below the sleep tolerance (0.0002 m/s = 0.005 units/s < b2_linearSleepTolerance): its tiny
    command re-wakes it every substep, both rest, the pair falls asleep at the end of substeps 5 and 10, and substep 10 is the one in
    which they stop touching (initial overlap 0.0044 units, 0.000463 units per substep)."""
    xy = np.array([[[0.0, 0.0], [0.8206 / 25.0, 0.0]]])
    th = np.array([[np.pi, 0.0]])
    a = np.zeros((1, 2, 2), np.float32)
    a[0, 0, 0] = 0.0002
    return xy, th, a


def test_a_contact_that_stops_touching_wakes_the_sleeping_partner():
    """b2Contact::Update: `if (touching != wasTouching) { bodyA->SetAwake(true); bodyB->SetAwake(true); }` -- ADVICE r02 (medium).
    In substep 11 the creeper wakes up (its command), finds the contact gone, and that wakes kilobot 1 although no island reaches it
    any more: its sleep time restarts (0.1 .. 0.4, asleep again at the end of substep 15).  Without the rule it would stay asleep."""
    xy, th, a = _creeper_scene()
    s = _sim(1, 2, xy[0], th=th, allow_sleep=1)
    s.set_actions(a)
    trace = []
    for k in range(20):
        s.step(1)
        trace.append(s.sleep_time[0].copy())
    trace = np.array(trace)
    assert (trace[4] < 0).all() and (trace[9] < 0).all()                       # asleep together at the end of substeps 5 and 10
    assert np.allclose(trace[10:14, 1], [0.1, 0.2, 0.3, 0.4], atol=1e-6)       # woken by the contact that ended
    assert trace[14, 1] < 0 and (trace[15:, 1] < 0).all()                      # ... asleep again, and left alone by the creeper from then on
    assert np.allclose(trace[15:19, 0], [0.1, 0.2, 0.3, 0.4], atol=1e-6)


@pytest.mark.gpu
def test_contact_end_wake_rule_on_the_device():
    from tests.test_parity_gpu import assert_same, assert_ws_same, dev, make_pair
    xy, th, a = _creeper_scene()
    E = 3                                                 # the same pair three times, padded with far-away kilobots in envs 1 and 2
    N = 8
    XY = np.zeros((E, N, 2)); TH = np.zeros((E, N)); A = np.zeros((E, N, 2), np.float32)
    XY[:, :, 0] = 0.3 + 0.08 * np.arange(N)[None]
    XY[:, :, 1] = 0.4
    for e in range(E):
        XY[e, 2 * e:2 * e + 2] = xy[0] + [0.1 * e, -0.2 * e]
        TH[e, 2 * e:2 * e + 2] = th[0]
        A[e, 2 * e] = a[0, 0]
    osim, gsim = make_pair(E, N, O.DRIVE_VELOCITY, O.LIGHT_NONE, xy=XY, th=TH, allow_sleep=1)
    osim.set_actions(A)
    gsim.set_actions(dev(A))
    woken = False
    for k in range(20):
        osim.step(1)
        gsim.step(1)
        assert_same(osim, gsim, 'substep %d' % k, SLEEP_FIELDS)
        woken = woken or (k == 11 and bool((osim.sleep_time[np.arange(E), 2 * np.arange(E) + 1] > 0).all()))
    assert woken
    assert_ws_same(osim, gsim, 'end')
