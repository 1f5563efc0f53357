"""A KilobotSim look-alike backed by the CPU oracle, so that the host-side env logic
(gym_kilobots_amd.envs / .lib) can be exercised without a GPU.  Lives in tests/ on purpose: the
product never routes through the oracle."""
import numpy as np
import torch

from oracle import oracle as O


class OracleBackend:
    def __init__(self, num_envs, num_bots, drive_mode=O.DRIVE_VELOCITY, light_type=O.LIGHT_NONE,
                 device=None, debug_outputs=False, **cfg):
        self.cfg = O.default_config(num_envs, num_bots, drive_mode, light_type, **cfg)
        self.o = O.OracleSim(self.cfg)
        self.num_envs, self.num_bots = num_envs, num_bots
        self.drive_mode, self.light_type = drive_mode, light_type
        for name in ('x', 'y', 'theta', 'v', 'w', 'acc_v', 'acc_w', 'motor_l', 'motor_r', 'pt_threshold',
                     'pt_update', 'pt_nochange', 'pt_dir', 'light_x', 'light_y', 'light_vx', 'light_vy', 'ws_cnt', 'status',
                     'light_value', 'light_gx', 'light_gy', 'cmd_vx', 'cmd_vy', 'cmd_w',
                     'ox', 'oy', 'otheta', 'ovx', 'ovy', 'ow', 'nbr_count', 'sleep_time', 'osleep', 'bot_mode'):
            arr = getattr(self.o, name)
            setattr(self, name, torch.from_numpy(arr.view(np.int32) if arr.dtype == np.uint32 else arr))     # shares memory
        if drive_mode not in (O.DRIVE_MOTORS, O.DRIVE_PHOTOTAXIS, O.DRIVE_MIXED):
            self.motor_l = self.motor_r = None
        if drive_mode != O.DRIVE_MIXED:
            self.bot_mode = None
        if not self.cfg.sense_radius > 0.0:
            self.nbr_count = None
        if not self.cfg.allow_sleep:
            self.sleep_time = self.osleep = None
        self.lds_bytes, self.block_threads = 0, 0
        self.device = torch.device('cpu')

    def sense(self, radius_m, out=None):
        return torch.from_numpy(self.o.sense(radius_m).view(np.int32))

    def light_sense(self, light_action=None):
        self.o.light_sense(None if light_action is None else light_action.cpu().numpy())

    def reset(self, **kw):
        self.o.reset(**kw)

    def status_bits(self):
        return int(np.bitwise_or.reduce(self.o.status)) if self.o.status.size else 0

    def host_state(self):
        M = self.cfg.num_objects
        objs = self.o.objects_m().astype(np.float32) if M else np.zeros((self.num_envs, 0, 3), np.float32)
        return self.o.poses_m().astype(np.float32), objs, self.o.status.view(np.int32).copy()

    def set_poses_m(self, xy, th):
        self.o.set_poses_m(xy, th)

    def forget_contacts(self):
        self.o.ws_cnt[...] = 0
        self.o.ows_acc[...] = -1.0

    def set_objects_m(self, xy, th=None):
        self.o.set_objects_m(xy, th)

    def object_poses(self):
        return torch.from_numpy(self.o.objects_m().astype(np.float32))

    def poses(self):
        return torch.from_numpy(self.o.poses_m().astype(np.float32))

    def set_actions(self, actions):
        self.o.set_actions(None if actions is None else actions.cpu().numpy())

    def step(self, n_substeps=1, actions=None, light_action=None, flags=0):
        if actions is not None:
            self.set_actions(actions)
        self.o.step(n_substeps, light_action=None if light_action is None else light_action.cpu().numpy(), flags=flags)

    def close(self):
        pass
