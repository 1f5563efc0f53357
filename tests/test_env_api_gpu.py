"""The env-level API on the real HIP path: same scenes as tests/test_env_api_cpu.py, compared with
the oracle-backed stand-in env step by step (bit-exact poses)."""
import numpy as np
import pytest
import torch

from gym_kilobots_amd.envs import DirectControlKilobotsEnv, KilobotsEnv, BatchedKilobotsEnv
from gym_kilobots_amd.lib import SimpleVelocityControlKilobot, PhototaxisKilobot, CircularGradientLight
from tests.oracle_backend import OracleBackend

pytestmark = pytest.mark.gpu


class CrowdEnv(DirectControlKilobotsEnv):
    def _configure_environment(self):
        rng = np.random.RandomState(0)
        for p in rng.normal(scale=0.05, size=(24, 2)):       # overlapping spawn, resolved by reset()
            self._add_kilobot(SimpleVelocityControlKilobot(self.world, position=p, orientation=rng.uniform(-3, 3),
                                                           velocity=[0.0, 0.0]))

    def get_reward(self, s, a, ns):
        return 0.


class LightEnv(KilobotsEnv):
    def _configure_environment(self):
        self._light = CircularGradientLight(position=np.array([0.2, 0.1]), radius=0.5)
        for i in range(15):
            self._add_kilobot(PhototaxisKilobot(self.world, position=(0.03 * (i % 5), 0.03 * (i // 5)), light=self._light))

    def get_reward(self, s, a, ns):
        return 1.


def test_direct_control_env_on_gpu_equals_oracle_env():
    g, o = CrowdEnv(), CrowdEnv(sim_factory=OracleBackend)
    og, oo = g.reset(), o.reset()
    assert np.array_equal(og['kilobots'], oo['kilobots'])
    rng = np.random.RandomState(1)
    for k in range(10):
        a = rng.uniform([0, -1.5], [0.01, 1.5], size=(24, 2))
        og, rg, dg, ig = g.step(a)
        oo, ro, do, io = o.step(a)
        assert np.array_equal(og['kilobots'], oo['kilobots']), k
    assert type(g.sim).__name__ == 'KilobotSim' and int(g.sim.status.max().item()) == 0
    kb = g.kilobots[3]
    np.testing.assert_allclose(kb.get_pose(), og['kilobots'][3], atol=1e-7)
    g.close()


def test_light_env_on_gpu_equals_oracle_env():
    g, o = LightEnv(), LightEnv(sim_factory=OracleBackend)
    g.reset(), o.reset()
    rng = np.random.RandomState(2)
    for k in range(8):
        a = rng.uniform(-0.02, 0.02, size=2)
        og, *_ = g.step(a)
        oo, *_ = o.step(a)
        assert np.array_equal(og['kilobots'], oo['kilobots']) and np.array_equal(og['light'], oo['light'])
    assert g.kilobots[0].get_motors() == o.kilobots[0].get_motors()


def test_batched_env_on_gpu():
    env = BatchedKilobotsEnv(32, 64, seed=5, reward_fn=lambda p, a, o: (o[..., :2] - p[..., :2]).norm(dim=-1).sum(-1))
    obs = env.reset()
    assert obs.is_cuda and obs.shape == (32, 64, 3)
    a = torch.rand(32, 64, 2, device=obs.device) * torch.tensor([0.01, 1.0], device=obs.device)
    for _ in range(3):
        obs, r, done, info = env.step(a)
    assert (r > 0).all() and torch.isfinite(obs).all()
    assert env.gather_episode_returns().shape == (32,)
