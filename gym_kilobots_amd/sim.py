"""KilobotSim: device buffers (torch-ROCm tensors) + the HIP world step behind the C ABI.

This is the host side of the hot path only.  torch is used for device memory and streams; all
simulation arithmetic happens inside libkilobots_hip.so (gym_kilobots_amd/csrc/).
"""
import ctypes as C

import numpy as np
import torch

from . import _native as nat
from ._native import STATUS_BITS, KilobotsStatusError, describe_status  # noqa: F401
from ._native import (DRIVE_VELOCITY, DRIVE_ACCEL, DRIVE_MOTORS, DRIVE_SIMPLE_PHOTOTAXIS,  # noqa: F401
                      DRIVE_PHOTOTAXIS, LIGHT_NONE, LIGHT_CIRCULAR, STEP_NO_DRIVE, WORLD_SCALE)


class KilobotSim:
    """num_envs independent worlds of num_bots kilobots, resident on one GPU.

    State tensors are [num_envs, num_bots] float32 in Box2D world units (metres x 25), exactly
    what the reference's b2Body objects hold; `poses()` returns metres like Body.get_pose
    (reference gym_kilobots/lib/body.py:63-65).
    """

    def __init__(self, num_envs, num_bots, drive_mode=DRIVE_VELOCITY, light_type=LIGHT_NONE,
                 device=None, debug_outputs=False, **cfg_overrides):
        self._lib = nat.load()                      # raises if the HIP library is not built
        if not torch.cuda.is_available():
            raise nat.KilobotsHipError('KilobotSim needs a ROCm GPU (torch.cuda.is_available() is False); '
                                       'there is no CPU fallback')
        self.device = torch.device(device if device is not None else 'cuda:%d' % torch.cuda.current_device())
        if self.device.index is None:
            self.device = torch.device('cuda:%d' % torch.cuda.current_device())
        self.cfg = nat.default_config(num_envs, num_bots, drive_mode, light_type, **cfg_overrides)
        self.num_envs, self.num_bots = num_envs, num_bots
        self.drive_mode, self.light_type = drive_mode, light_type
        self._h = C.c_void_p()
        nat.check(self._lib.kb_create(C.byref(self.cfg), C.byref(self._h)), 'kb_create')
        E, N = num_envs, num_bots
        dev = self.device
        f = lambda *s: torch.zeros(*s, dtype=torch.float32, device=dev)
        self.x, self.y, self.theta = f(E, N), f(E, N), f(E, N)
        self.v = self.w = self.acc_v = self.acc_w = None
        self.motor_l = self.motor_r = None
        self.pt_threshold = self.pt_update = self.pt_nochange = self.pt_dir = None
        self.light_x = self.light_y = None
        mixed = drive_mode == nat.DRIVE_MIXED      # every per-law state buffer + the per-kilobot law
        self.bot_mode = torch.full((E, N), DRIVE_MOTORS, dtype=torch.uint8, device=dev) if mixed else None
        if drive_mode in (DRIVE_VELOCITY, DRIVE_ACCEL) or mixed:
            self.v, self.w = f(E, N), f(E, N)
        if drive_mode == DRIVE_ACCEL or mixed:
            self.acc_v, self.acc_w = f(E, N), f(E, N)
        if drive_mode in (DRIVE_MOTORS, DRIVE_PHOTOTAXIS) or mixed:
            # Kilobot._setup -> turn_left (kilobot.py:78-81, 315-316)
            self.motor_l = torch.full((E, N), 255, dtype=torch.uint8, device=dev)
            self.motor_r = torch.zeros(E, N, dtype=torch.uint8, device=dev)
        if drive_mode == DRIVE_PHOTOTAXIS or mixed:
            self.pt_threshold = torch.full((E, N), float('-inf'), dtype=torch.float32, device=dev)
            self.pt_update = torch.zeros(E, N, dtype=torch.int32, device=dev)
            self.pt_nochange = torch.zeros(E, N, dtype=torch.int32, device=dev)
            self.pt_dir = torch.zeros(E, N, dtype=torch.uint8, device=dev)
        self.light_vx = self.light_vy = None
        if light_type != LIGHT_NONE:
            LC = self._lib.kb_light_count(self._h)
            shape = (E,) if LC == 1 else (E, LC)
            self.light_x, self.light_y, self.light_vx, self.light_vy = f(*shape), f(*shape), f(*shape), f(*shape)
        cap = self._lib.kb_contact_capacity(self._h)
        self.ws_key = torch.zeros(E, cap, dtype=torch.int32, device=dev)
        self.ws_acc = f(E, cap)
        self.ws_cnt = torch.zeros(E, N, dtype=torch.uint8, device=dev)
        self.scratch = torch.empty(self._lib.kb_scratch_bytes(self._h), dtype=torch.uint8, device=dev)
        self.status = torch.zeros(E, dtype=torch.int32, device=dev)
        self._status_mask = None
        self.num_objects = M = self.cfg.num_objects
        self.ox = self.oy = self.otheta = self.ovx = self.ovy = self.ow = self.ows_acc = None
        if M > 0:
            self.ox, self.oy, self.otheta = f(E, M), f(E, M), f(E, M)
            self.ovx, self.ovy, self.ow = f(E, M), f(E, M), f(E, M)
            self.ows_acc = torch.full((E, nat.MAX_OBJECTS, nat.OWS_COLS, nat.OWS_WORDS), -1.0, dtype=torch.float32, device=dev)
        # sleeping (kb_config.allow_sleep): b2Body::m_sleepTime of every kilobot / object, < 0 = asleep
        self.sleep_time = self.osleep = None
        if self.cfg.allow_sleep:
            self.sleep_time = f(E, N)
            if M > 0:
                self.osleep = f(E, M)
        # IR-range neighbour sensing (kb_config.sense_radius): counts of the last substep's sensing point
        self.nbr_count = None
        if self.cfg.sense_radius > 0.0:
            self.nbr_count = torch.zeros(E, N, dtype=torch.int32, device=dev)
        self.light_value = self.light_gx = self.light_gy = None
        self.cmd_vx = self.cmd_vy = self.cmd_w = None
        if debug_outputs:
            self.light_value, self.light_gx, self.light_gy = f(E, N), f(E, N), f(E, N)
            self.cmd_vx, self.cmd_vy, self.cmd_w = f(E, N), f(E, N), f(E, N)
        self._bind()

    # ------------------------------------------------------------------ plumbing
    def _bind(self):
        b = nat.KbBuffers()
        for name in nat.BUFFER_FIELDS:
            t = getattr(self, name, None)
            setattr(b, name, None if t is None else t.data_ptr())
        nat.check(self._lib.kb_bind(self._h, C.byref(b)), 'kb_bind')

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def close(self):
        if getattr(self, '_h', None) is not None and self._h:
            self._lib.kb_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def lds_bytes(self):
        return self._lib.kb_lds_bytes(self._h)

    @property
    def resident_envs_per_cu(self):
        """Workgroups (= envs) of this sim's kernel that one CU holds at a time (HIP occupancy query)."""
        with torch.cuda.device(self.device):
            n = self._lib.kb_resident_envs_per_cu(self._h)
        if n < 0:
            nat.check(n, 'kb_resident_envs_per_cu')
        return n

    @property
    def contact_capacity(self):
        return self._lib.kb_contact_capacity(self._h)

    @property
    def lds_staging_entries(self):
        """Contacts of one env that are staged in LDS (an env with more takes the global staging slice for that substep)."""
        return self._lib.kb_lds_staging_entries(self._h)

    @property
    def block_threads(self):
        return self._lib.kb_block_threads(self._h)

    @block_threads.setter
    def block_threads(self, n):
        nat.check(self._lib.kb_set_block_threads(self._h, int(n)), 'kb_set_block_threads')

    # ------------------------------------------------------------------ state
    def set_poses_m(self, xy_m, theta):
        """Body poses in metres / radians; stored as fp32 world units like body.py:32-34 does."""
        xy = np.asarray(xy_m, np.float64) * WORLD_SCALE
        self.x.copy_(torch.from_numpy(np.ascontiguousarray(xy[..., 0].astype(np.float32))).reshape(self.x.shape))
        self.y.copy_(torch.from_numpy(np.ascontiguousarray(xy[..., 1].astype(np.float32))).reshape(self.y.shape))
        self.theta.copy_(torch.from_numpy(np.ascontiguousarray(np.asarray(theta, np.float32))).reshape(self.theta.shape))
        if self.sleep_time is not None:
            self.sleep_time.zero_()          # re-created bodies are awake
        self.forget_contacts()

    def forget_contacts(self):
        """Drop all warm-start impulses (bodies were re-created / teleported)."""
        self.ws_cnt.zero_()
        if self.ows_acc is not None:
            self.ows_acc.fill_(-1.0)

    def set_objects_m(self, xy_m, theta=None):
        """Object poses in metres / radians; objects start at rest (Body.__init__, body.py:32-38)."""
        xy = np.asarray(xy_m, np.float64) * WORLD_SCALE
        self.ox.copy_(torch.from_numpy(np.ascontiguousarray(xy[..., 0].astype(np.float32))).reshape(self.ox.shape))
        self.oy.copy_(torch.from_numpy(np.ascontiguousarray(xy[..., 1].astype(np.float32))).reshape(self.oy.shape))
        if theta is None:
            self.otheta.zero_()
        else:
            self.otheta.copy_(torch.from_numpy(np.ascontiguousarray(np.asarray(theta, np.float32))).reshape(self.otheta.shape))
        self.ovx.zero_()
        self.ovy.zero_()
        self.ow.zero_()
        if self.osleep is not None:
            self.osleep.zero_()
        self.forget_contacts()

    def object_poses(self):
        """[num_envs, num_objects, 3] float32 (x [m], y [m], theta): get_state()['objects'] of every env."""
        # tensor / tensor: a true IEEE division like Body.get_pose (torch multiplies by the reciprocal for tensor / scalar)
        scale = torch.full_like(self.ox, WORLD_SCALE)
        return torch.stack([torch.div(self.ox, scale), torch.div(self.oy, scale), self.otheta], -1)

    def poses(self):
        """[num_envs, num_bots, 3] float32 (x [m], y [m], theta): get_state()['kilobots'] of every env."""
        out = torch.empty(self.num_envs, self.num_bots, 3, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            nat.check(self._lib.kb_get_poses(self._h, C.c_void_p(out.data_ptr()), self._stream()), 'kb_get_poses')
        return out

    def host_state(self):
        """(kilobot poses [num_envs, num_bots, 3], object poses [num_envs, num_objects, 3], status [num_envs]) as numpy
        arrays, metres / radians: one launch and ONE copy to the host (kb_get_state), for get_state() after every step."""
        N, M = self.num_bots, self.num_objects
        out = torch.empty(self.num_envs, 3 * (N + M) + 1, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            nat.check(self._lib.kb_get_state(self._h, C.c_void_p(out.data_ptr()), self._stream()), 'kb_get_state')
        h = out.cpu().numpy()
        E = self.num_envs
        return (h[:, :3 * N].reshape(E, N, 3), h[:, 3 * N:3 * (N + M)].reshape(E, M, 3),
                np.ascontiguousarray(h[:, -1]).view(np.int32))

    def status_bits(self):
        """OR of the status flags of all envs (one device read; synchronises the stream)."""
        s = self.status
        if s.numel() <= (1 << 16):           # one small copy; OR on the host (4 reductions + 4 syncs cost 0.1 ms per env.step)
            return int(np.bitwise_or.reduce(s.cpu().numpy(), initial=0)) & sum(STATUS_BITS)
        if self._status_mask is None:
            self._status_mask = torch.tensor(sorted(STATUS_BITS), dtype=torch.int32, device=s.device)
        hit = (s.unsqueeze(-1) & self._status_mask).ne(0).any(0).cpu().numpy()
        return int(sum(b for b, h in zip(sorted(STATUS_BITS), hit) if h))

    def check_status(self, mode='raise', where=''):
        """Surface capacity overflows of the device step: mode 'raise' | 'warn' | 'ignore'.  Returns the bits."""
        if mode == 'ignore':
            return 0
        bits = self.status_bits()
        if bits:
            bad = int((self.status != 0).sum().item())
            msg = '%sdevice step status 0x%x in %d of %d envs: %s' % (where and where + ': ', bits, bad, self.num_envs, describe_status(bits))
            if mode == 'raise':
                raise KilobotsStatusError(msg)
            import warnings
            warnings.warn(msg, RuntimeWarning, stacklevel=3)
        return bits

    def sense(self, radius_m, out=None):
        """IR-range neighbour sensing on the current poses: [num_envs, num_bots] int32 counts of the kilobots within
        radius_m (centre to centre) of each kilobot (kb_sense; no reference counterpart)."""
        if out is None:
            out = torch.empty(self.num_envs, self.num_bots, dtype=torch.int32, device=self.device)
        if not (out.is_cuda and out.dtype == torch.int32 and out.is_contiguous() and tuple(out.shape) == (self.num_envs, self.num_bots)):
            raise ValueError('out must be a contiguous int32 cuda tensor of shape (num_envs, num_bots)')
        with torch.cuda.device(self.device):
            nat.check(self._lib.kb_sense(self._h, float(radius_m), C.c_void_p(out.data_ptr()), self._stream()), 'kb_sense')
        return out

    def light_sense(self, light_action=None):
        """The sensing point of one substep on its own (kb_light_sense): Light.step with `light_action` (None: the light
        stays) + value_and_gradients at every kilobot's sensor into light_value / light_gx / light_gy -- for kilobots whose
        _loop runs on the host between the sensing and the motor law (kilobots_env.py:171-184)."""
        if self.light_value is None:
            raise ValueError('light_sense needs the sensing outputs: create the sim with debug_outputs=True')
        pl = self._ptr(light_action, (self.num_envs, self._lib.kb_light_action_dim(self._h)), 'light_action')
        with torch.cuda.device(self.device):
            nat.check(self._lib.kb_light_sense(self._h, pl, self._stream()), 'kb_light_sense')

    def reset(self, seed=0, mean=(0.0, 0.0), std=0.1, random_theta=False, random_velocity=False, resolve=True, env_offset=0):
        """KilobotsEnv.reset of every env on the device (kb_reset): Gaussian spawn clipped to the bounds -/+ 0.02 m
        (yaml_kilobots_env.py:346-352), Philox4x32-10 keyed by (seed; env_offset + env, bot), then the step to resolve."""
        rp = nat.KbResetParams()
        rp.seed, rp.env_offset = int(seed) & 0xFFFFFFFFFFFFFFFF, int(env_offset)
        rp.mean[0], rp.mean[1], rp.std = float(mean[0]), float(mean[1]), float(std)
        rp.random_theta, rp.random_velocity, rp.resolve = int(bool(random_theta)), int(bool(random_velocity)), int(bool(resolve))
        with torch.cuda.device(self.device):
            nat.check(self._lib.kb_reset(self._h, C.byref(rp), self._stream()), 'kb_reset')

    # ------------------------------------------------------------------ stepping
    def _ptr(self, t, shape, name):
        if t is None:
            return None
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and tuple(t.shape) == tuple(shape)):
            raise ValueError('%s must be a contiguous float32 cuda tensor of shape %s' % (name, tuple(shape)))
        if t.device != self.device:
            raise ValueError('%s lives on %s, the simulator on %s' % (name, t.device, self.device))
        return C.c_void_p(t.data_ptr())

    def set_actions(self, actions):
        """set_action of every kilobot (clamped); actions [E, N, 2] cuda float32 or None (= zeros)."""
        p = self._ptr(actions, (self.num_envs, self.num_bots, 2), 'actions')
        with torch.cuda.device(self.device):    # the launch goes to the CURRENT HIP device: make it the sim's
            nat.check(self._lib.kb_set_actions(self._h, p, self._stream()), 'kb_set_actions')

    def step(self, n_substeps=1, actions=None, light_action=None, flags=0):
        """n_substeps iterations of the reference substep loop in one kernel launch (asynchronous)."""
        pa = self._ptr(actions, (self.num_envs, self.num_bots, 2), 'actions')
        pl = self._ptr(light_action, (self.num_envs, self._lib.kb_light_action_dim(self._h)), 'light_action')
        with torch.cuda.device(self.device):
            nat.check(self._lib.kb_step(self._h, pa, pl, int(n_substeps), int(flags), self._stream()), 'kb_step')
