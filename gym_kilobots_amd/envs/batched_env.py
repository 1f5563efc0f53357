"""BatchedKilobotsEnv: the tensor-level API for many envs on one GPU (what RL training loops use).

Same semantics as DirectControlKilobotsEnv / KilobotsEnv per env, but observations stay on the
device as torch tensors: poses [E, N, 3] (metres, radians), actions [E, N, 2]."""
import numpy as np
import torch

from .. import _native as nat
from ..lib.kilobot import Kilobot
from ..spaces import Box


class BatchedKilobotsEnv(object):
    steps_per_action = 10            # kilobots_env.py:28
    sim_step = 0.1                   # kilobots_env.py:25,32

    def __init__(self, num_envs, num_kilobots, drive_mode=nat.DRIVE_VELOCITY, light_type=nat.LIGHT_NONE,
                 world_size=(2.0, 1.5), spawn_std=0.1, spawn_mean=(0.0, 0.0), seed=0, device=None,
                 sim_factory=None, reward_fn=None, **cfg):
        if sim_factory is None:
            from ..sim import KilobotSim as sim_factory
        self.num_envs, self.num_kilobots = int(num_envs), int(num_kilobots)
        self.world_width, self.world_height = world_size
        self.spawn_std, self.spawn_mean = spawn_std, np.asarray(spawn_mean, dtype=np.float64)
        self._rng = np.random.RandomState(seed)
        self._seed = seed
        self.reward_fn = reward_fn
        kw = dict(cfg)
        if device is not None:
            kw['device'] = device
        self.sim = sim_factory(self.num_envs, self.num_kilobots, drive_mode, light_type,
                               world_width=self.world_width, world_height=self.world_height, **kw)
        lo = np.array([0.0, -Kilobot._max_angular_velocity])
        hi = np.array([Kilobot._max_linear_velocity, Kilobot._max_angular_velocity])
        if drive_mode == nat.DRIVE_ACCEL:
            lo, hi = np.array([-.005, -.2 * np.pi]), np.array([.005, .2 * np.pi])
        self.action_space = Box(np.tile(lo, (self.num_kilobots, 1)), np.tile(hi, (self.num_kilobots, 1)), dtype=np.float64)
        b = np.array([self.world_width / 2, self.world_height / 2, np.inf])
        self.observation_space = Box(np.tile(-b, (self.num_kilobots, 1)), np.tile(b, (self.num_kilobots, 1)), dtype=np.float32)
        self.episode_returns = torch.zeros(self.num_envs, dtype=torch.float32, device=self.sim.x.device)

    def seed(self, seed=None):
        if seed is not None:
            self._seed = seed
            self._rng = np.random.RandomState(seed)
        return [self._seed]

    def spawn(self):
        """YamlKilobotsEnv._init_kilobots spawn rule (yaml_kilobots_env.py:346-352), theta = 0 (body.py:28-29)."""
        E, N = self.num_envs, self.num_kilobots
        xy = self._rng.normal(scale=self.spawn_std, size=(E, N, 2)) + self.spawn_mean
        lo = np.array([-self.world_width / 2, -self.world_height / 2]) + 0.02
        hi = np.array([self.world_width / 2, self.world_height / 2]) - 0.02
        return np.minimum(np.maximum(xy, lo), hi), np.zeros((E, N))

    def reset(self, poses=None):
        """poses: optional (xy [E,N,2] metres, theta [E,N]); default: the reference's Gaussian spawn."""
        xy, th = self.spawn() if poses is None else poses
        self.sim.set_poses_m(xy, th)
        self.sim.status.zero_()
        self.episode_returns.zero_()
        self.sim.step(1, flags=nat.STEP_NO_DRIVE)       # "step to resolve", kilobots_env.py:156-157
        return self.sim.poses()

    def step(self, actions=None, light_action=None):
        """actions [E, N, 2] float32 tensor on the sim's device (None keeps the previous commands)."""
        prev = self.sim.poses() if self.reward_fn is not None else None
        self.sim.step(self.steps_per_action, actions=actions, light_action=light_action)
        obs = self.sim.poses()
        if self.reward_fn is not None:
            reward = self.reward_fn(prev, actions, obs)
        else:
            reward = torch.zeros(self.num_envs, dtype=torch.float32, device=obs.device)
        self.episode_returns += reward
        done = torch.zeros(self.num_envs, dtype=torch.bool, device=obs.device)
        return obs, reward, done, {}

    def gather_episode_returns(self, dist=None):
        """Per-env returns of every rank's shard in global env order (the only collective)."""
        from ..dist import gather_returns
        return gather_returns(self.episode_returns, dist)

    def close(self):
        self.sim.close()
