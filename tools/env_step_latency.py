"""Wall time of the drop-in single-env API (gym-style KilobotsEnv.step, host numpy in / out) on the HIP path.
    python tools/env_step_latency.py [num_kilobots]"""
import sys, time
import numpy as np
sys.path.insert(0, '.')
from gym_kilobots_amd.envs import DirectControlKilobotsEnv
from gym_kilobots_amd.lib import SimpleVelocityControlKilobot, Quad

N = int(sys.argv[1]) if len(sys.argv) > 1 else 16


class Env(DirectControlKilobotsEnv):
    def _configure_environment(self):
        rng = np.random.RandomState(0)
        self._add_object(Quad(world=self.world, width=.15, height=.15, position=(0.3, 0.0), orientation=0.2))
        for p in rng.uniform(-0.3, 0.1, size=(N, 2)):
            self._add_kilobot(SimpleVelocityControlKilobot(self.world, position=p, orientation=rng.uniform(-3, 3), velocity=[0.0, 0.0]))

    def get_reward(self, s, a, ns):
        return 0.


for sleep in (True, False):
    env = Env(allow_sleep=sleep)
    env.reset()
    a = np.tile(np.array([[0.01, 0.2]]), (N, 1))
    for _ in range(20):
        env.step(a)
    t0 = time.perf_counter()
    K = 300
    for _ in range(K):
        obs, r, d, info = env.step(a)
    dt = (time.perf_counter() - t0) / K
    print('KilobotsEnv.step with %d kilobots + 1 box, allow_sleep=%s: %.3f ms per env.step (10 substeps), %.0f env.steps/s' % (N, sleep, dt * 1e3, 1.0 / dt))
    import torch
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        env.sim.step(10)
    torch.cuda.synchronize()
    print('   the launch alone (sim.step(10), no host round trip): %.3f ms' % ((time.perf_counter() - t0) / K * 1e3))
    env.close()
