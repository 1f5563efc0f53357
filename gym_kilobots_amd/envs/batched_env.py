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
                 sim_factory=None, reward_fn=None, env_offset=0, on_status='raise', status_interval=1, **cfg):
        """env_offset: global index of this shard's first env (multi-GPU: the Philox counters of reset() are keyed by the
        GLOBAL env index, so a shard equals the corresponding rows of the unsharded batch).
        on_status / status_interval: capacity overflows of the device step (kb_buffers.status) are checked after
        reset() and after every status_interval-th step(): 'raise' | 'warn' | 'ignore' (one 4-byte device read each)."""
        if sim_factory is None:
            from ..sim import KilobotSim as sim_factory
        if on_status not in ('raise', 'warn', 'ignore'):
            raise ValueError("on_status must be 'raise', 'warn' or 'ignore'")
        self.num_envs, self.num_kilobots = int(num_envs), int(num_kilobots)
        self.world_width, self.world_height = world_size
        self.spawn_std, self.spawn_mean = spawn_std, np.asarray(spawn_mean, dtype=np.float64)
        if seed is None:        # (gym convention: no seed given -> draw one; the device reset needs an integer key)
            seed = int(np.random.SeedSequence().entropy) & 0x7FFFFFFF
        self._rng = np.random.RandomState(seed)
        self._seed = seed
        self._resets = 0
        self.env_offset = int(env_offset)
        self._on_status, self._status_interval, self._steps = on_status, max(1, int(status_interval)), 0
        self.reward_fn = reward_fn
        kw = dict(cfg)
        if 'contact_capacity' not in kw:
            # a Gaussian cloud of std s overlaps N (N - 1) / 2 * (1 - exp(-r^2 / s^2)) pairs at spawn: size the contact
            # store for it (entries live in HBM), so that the reference's default spawn never drops contacts silently
            r, n = float(kw.get('bot_radius', 0.0165)), self.num_kilobots
            pairs = 0.5 * n * (n - 1) * (1.0 - np.exp(-(r * r) / max(float(spawn_std) ** 2, 1e-12)))
            default_cap = max(4 * n + 64, min(n * (n - 1) // 2 + 4 * n, 2304))
            need = int(1.5 * pairs) + 4 * n + 64
            if need > default_cap:
                kw['contact_capacity'] = min(need, n * (n - 1) // 2 + 4 * n, 65528)
                kw.setdefault('ws_slots', 64)
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
            self._resets = 0
        return [self._seed]

    def _check_status(self, where):
        if self._on_status == 'ignore':
            return
        bits = self.sim.status_bits()
        if not bits:
            return
        msg = '%s: device step status 0x%x: %s' % (where, bits, nat.describe_status(bits))
        if self._on_status == 'raise':
            raise nat.KilobotsStatusError(msg)
        import warnings
        warnings.warn(msg, RuntimeWarning, stacklevel=3)
        self.sim.status.zero_()

    def spawn(self):
        """YamlKilobotsEnv._init_kilobots spawn rule (yaml_kilobots_env.py:346-352), theta = 0 (body.py:28-29)."""
        E, N = self.num_envs, self.num_kilobots
        xy = self._rng.normal(scale=self.spawn_std, size=(E, N, 2)) + self.spawn_mean
        lo = np.array([-self.world_width / 2, -self.world_height / 2]) + 0.02
        hi = np.array([self.world_width / 2, self.world_height / 2]) - 0.02
        return np.minimum(np.maximum(xy, lo), hi), np.zeros((E, N))

    def reset(self, poses=None):
        """poses: optional (xy [E,N,2] metres, theta [E,N]) uploaded from the host.  Default: the reference's Gaussian
        spawn drawn ON THE DEVICE (kb_reset: Philox4x32-10 keyed by (seed + reset count; global env, bot)), no host
        arrays, no H2D copy; followed by the "step to resolve" of kilobots_env.py:156-157 either way."""
        if poses is None:
            self.sim.reset(seed=(int(self._seed) << 20) + self._resets, mean=tuple(float(v) for v in self.spawn_mean),
                           std=float(self.spawn_std), random_theta=False, random_velocity=False, resolve=True,
                           env_offset=self.env_offset)
            self._resets += 1
        else:
            xy, th = poses
            self.sim.set_poses_m(xy, th)
            self.sim.status.zero_()
            self.sim.step(1, flags=nat.STEP_NO_DRIVE)       # "step to resolve", kilobots_env.py:156-157
        self.episode_returns.zero_()
        self._steps = 0
        self._check_status('reset()')
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
        self._steps += 1
        if self._steps % self._status_interval == 0:
            self._check_status('step()')
        done = torch.zeros(self.num_envs, dtype=torch.bool, device=obs.device)
        return obs, reward, done, {}

    def gather_episode_returns(self, dist=None):
        """Per-env returns of every rank's shard in global env order (the only collective)."""
        from ..dist import gather_returns
        return gather_returns(self.episode_returns, dist)

    def close(self):
        self.sim.close()
