"""ctypes wrapper of the CPU ORACLE (oracle/kb_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg as the checker; the product package gym_kilobots_amd never imports it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, 'libkb_oracle.so')

DRIVE_VELOCITY, DRIVE_ACCEL, DRIVE_MOTORS, DRIVE_SIMPLE_PHOTOTAXIS, DRIVE_PHOTOTAXIS, DRIVE_MIXED = range(6)
LIGHT_NONE, LIGHT_CIRCULAR, LIGHT_GRADIENT, LIGHT_MOMENTUM, LIGHT_COMPOSITE = range(5)
MAX_LIGHTS = 4
STEP_NO_DRIVE = 1
MAX_OBJECTS = 8
MAX_POLY_VERTS = 4
SHAPE_CIRCLE, SHAPE_BOX, SHAPE_POLYGON = range(3)
OWS_COLS, OWS_WORDS = 12, 6
WORLD_SCALE = 25.0


class Config(C.Structure):
    _fields_ = [
        ('num_envs', C.c_int32), ('num_bots', C.c_int32), ('num_objects', C.c_int32),
        ('world_width', C.c_float), ('world_height', C.c_float),
        ('dt', C.c_float), ('vel_iters', C.c_int32), ('pos_iters', C.c_int32),
        ('drive_mode', C.c_int32), ('light_type', C.c_int32),
        ('bot_radius', C.c_float), ('bot_density', C.c_float),
        ('bot_linear_damping', C.c_float), ('bot_angular_damping', C.c_float),
        ('light_radius', C.c_float),
        ('light_lo', C.c_float * 2), ('light_hi', C.c_float * 2),
        ('light_act_lo', C.c_float * 2), ('light_act_hi', C.c_float * 2),
        ('light_max_velocity', C.c_float),
        ('ws_slots', C.c_int32),
        ('obj_radius', C.c_float * MAX_OBJECTS),
        ('obj_density', C.c_float), ('obj_friction', C.c_float),
        ('obj_linear_damping', C.c_float), ('obj_angular_damping', C.c_float),
        ('toi_walls', C.c_int32),
        ('solver_mode', C.c_int32),
        ('light_count', C.c_int32), ('light_kind', C.c_int32 * MAX_LIGHTS),
        ('lightc_radius', C.c_float * MAX_LIGHTS), ('lightc_max_velocity', C.c_float * MAX_LIGHTS),
        ('lightc_lo', (C.c_float * 2) * MAX_LIGHTS), ('lightc_hi', (C.c_float * 2) * MAX_LIGHTS),
        ('lightc_act_lo', (C.c_float * 2) * MAX_LIGHTS), ('lightc_act_hi', (C.c_float * 2) * MAX_LIGHTS),
        ('obj_shape', C.c_int32 * MAX_OBJECTS), ('obj_nverts', C.c_int32 * MAX_OBJECTS),
        ('obj_verts', ((C.c_float * 2) * MAX_POLY_VERTS) * MAX_OBJECTS),
        ('wall_friction', C.c_float),
        ('num_fixtures', C.c_int32), ('obj_fixture_body', C.c_int32 * MAX_OBJECTS),
        ('damping_model', C.c_int32), ('sense_radius', C.c_float), ('contact_capacity', C.c_int32),
        ('mode_density', C.c_float * 5),
        ('allow_sleep', C.c_int32),
    ]


class ResetParams(C.Structure):
    _fields_ = [('seed', C.c_uint64), ('env_offset', C.c_int32), ('mean', C.c_float * 2), ('std', C.c_float),
                ('random_theta', C.c_int32), ('random_velocity', C.c_int32), ('resolve', C.c_int32)]


_PF = C.POINTER(C.c_float)
_PU8 = C.POINTER(C.c_uint8)
_PI32 = C.POINTER(C.c_int32)
_PU32 = C.POINTER(C.c_uint32)


class State(C.Structure):
    _fields_ = [
        ('x', _PF), ('y', _PF), ('theta', _PF), ('v', _PF), ('w', _PF),
        ('acc_v', _PF), ('acc_w', _PF), ('motor_l', _PU8), ('motor_r', _PU8),
        ('pt_threshold', _PF), ('pt_update', _PI32), ('pt_nochange', _PI32), ('pt_dir', _PU8),
        ('light_x', _PF), ('light_y', _PF), ('light_vx', _PF), ('light_vy', _PF),
        ('ox', _PF), ('oy', _PF), ('otheta', _PF), ('ovx', _PF), ('ovy', _PF), ('ow', _PF),
        ('ws_key', _PU32), ('ws_acc', _PF), ('ws_cnt', _PU8),
        ('light_value', _PF), ('light_gx', _PF), ('light_gy', _PF),
        ('cmd_vx', _PF), ('cmd_vy', _PF), ('cmd_w', _PF),
        ('status', _PI32),
        ('ows_acc', _PF),
        ('nbr_count', _PU32),
        ('sleep_time', _PF), ('osleep', _PF),
        ('bot_mode', _PU8),
    ]


def build(force=False):
    """Compile oracle/libkb_oracle.so with gcc (called by __graft_entry__.build and tests)."""
    src = os.path.join(_HERE, 'kb_oracle.c')
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < max(
            os.path.getmtime(src), os.path.getmtime(os.path.join(_HERE, 'kb_oracle.h'))):
        subprocess.check_call(['make', '-C', _HERE, '-B', 'libkb_oracle.so'], stdout=subprocess.DEVNULL)
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB)
        _lib.kbo_step.argtypes = [C.POINTER(Config), C.POINTER(State), _PF, C.c_int, C.c_int, C.c_int]
        _lib.kbo_step.restype = C.c_int
        _lib.kbo_set_actions.argtypes = [C.POINTER(Config), C.POINTER(State), _PF]
        _lib.kbo_set_actions.restype = C.c_int
        _lib.kbo_count_contacts.argtypes = [C.POINTER(Config), C.POINTER(State), C.c_int, _PI32, _PI32, _PI32]
        _lib.kbo_count_contacts.restype = C.c_int
        _lib.kbo_contact_capacity.argtypes = [C.POINTER(Config)]
        _lib.kbo_contact_capacity.restype = C.c_int
        _lib.kbo_sincosf.argtypes = [C.c_float, _PF, _PF]
        _lib.kbo_sincosf.restype = None
        _lib.kbo_logf.argtypes = [C.c_float]
        _lib.kbo_logf.restype = C.c_float
        _lib.kbo_sense.argtypes = [C.POINTER(Config), C.POINTER(State), C.c_float, _PU32]
        _lib.kbo_sense.restype = C.c_int
        _lib.kbo_light_sense.argtypes = [C.POINTER(Config), C.POINTER(State), _PF]
        _lib.kbo_light_sense.restype = C.c_int
        _lib.kbo_philox4x32_10.argtypes = [_PU32, _PU32, _PU32]
        _lib.kbo_philox4x32_10.restype = None
        _lib.kbo_reset.argtypes = [C.POINTER(Config), C.POINTER(State), C.POINTER(ResetParams)]
        _lib.kbo_reset.restype = C.c_int
    return _lib


def default_config(num_envs, num_bots, drive_mode=DRIVE_VELOCITY, light_type=LIGHT_NONE, **kw):
    """Reference defaults: kilobots_env.py:19,25-28; kilobot.py:9,25-30,214; body.py:11-16."""
    c = Config()
    c.num_envs, c.num_bots, c.num_objects = num_envs, num_bots, 0
    for i in range(MAX_OBJECTS):
        c.obj_radius[i] = 0.075
    c.world_width, c.world_height = 2.0, 1.5
    c.dt, c.vel_iters, c.pos_iters = 0.1, 10, 10
    c.drive_mode, c.light_type = drive_mode, light_type
    c.bot_radius = 0.0165
    c.bot_density = 2.0 if drive_mode in (DRIVE_VELOCITY, DRIVE_ACCEL) else 1.0
    c.bot_linear_damping = c.bot_angular_damping = 0.8
    c.light_radius = 0.2
    c.light_lo[0] = c.light_lo[1] = -np.inf
    c.light_hi[0] = c.light_hi[1] = np.inf
    c.light_act_lo[0] = c.light_act_lo[1] = -0.01
    c.light_act_hi[0] = c.light_act_hi[1] = 0.01
    c.light_max_velocity = np.inf
    c.ws_slots = 32     # contacts per kilobot whose impulse is carried over (Box2D keeps every b2Contact; 32 covers a dense overlapping spawn)
    c.obj_density, c.obj_friction = 2.0, 0.01
    c.obj_linear_damping = c.obj_angular_damping = 0.8
    c.toi_walls = 1      # b2World continuousPhysics defaults to true
    c.wall_friction = 0.2  # b2FixtureDef default (the arena chain, kilobots_env.py:46-51)
    c.damping_model = 0    # Pade (Box2D >= 2.3.1)
    c.sense_radius = 0.0
    c.contact_capacity = 0
    c.allow_sleep = 1      # b2World(gravity=(0, 0), doSleep=True) of kilobots_env.py:45, like the product default
    c.light_count = 1
    for i in range(MAX_LIGHTS):
        c.light_kind[i] = LIGHT_CIRCULAR
        c.lightc_radius[i] = 0.2
        c.lightc_max_velocity[i] = np.inf
        for k in range(2):
            c.lightc_lo[i][k], c.lightc_hi[i][k] = -np.inf, np.inf
            c.lightc_act_lo[i][k], c.lightc_act_hi[i][k] = -0.01, 0.01
    for k, v in kw.items():
        _assign(c, k, v)
    return c


def _assign(c, k, v):
    cur = getattr(c, k)
    if hasattr(cur, '__len__'):
        _fill(cur, v)
    else:
        setattr(c, k, v)


def _fill(dst, src):
    for i, vi in enumerate(src):
        if hasattr(dst[i], '__len__'):
            _fill(dst[i], vi)
        else:
            dst[i] = vi


class OracleSim:
    """State container + stepping; numpy arrays are [num_envs, num_bots] float32 in world units."""

    def __init__(self, cfg):
        self.cfg = cfg
        E, N = cfg.num_envs, cfg.num_bots
        self.cap = lib().kbo_contact_capacity(C.byref(cfg))
        f = lambda *s: np.zeros(s, np.float32)
        self.x, self.y, self.theta, self.v, self.w = f(E, N), f(E, N), f(E, N), f(E, N), f(E, N)
        self.acc_v, self.acc_w = f(E, N), f(E, N)
        self.motor_l = np.full((E, N), 255, np.uint8)     # Kilobot._setup -> turn_left, kilobot.py:78-81,315-316
        self.motor_r = np.zeros((E, N), np.uint8)
        self.pt_threshold = np.full((E, N), -np.inf, np.float32)
        self.pt_update = np.zeros((E, N), np.int32)
        self.pt_nochange = np.zeros((E, N), np.int32)
        self.pt_dir = np.zeros((E, N), np.uint8)
        LC = cfg.light_count if cfg.light_type == LIGHT_COMPOSITE else 1
        self.light_x, self.light_y, self.light_vx, self.light_vy = f(E, LC), f(E, LC), f(E, LC), f(E, LC)
        if LC == 1:
            self.light_x, self.light_y = self.light_x.reshape(E), self.light_y.reshape(E)
            self.light_vx, self.light_vy = self.light_vx.reshape(E), self.light_vy.reshape(E)
        self.ws_key = np.full((E, self.cap), 0xFFFFFFFF, np.uint32)
        self.ws_acc = f(E, self.cap)
        self.ws_cnt = np.zeros((E, N), np.uint8)
        self.light_value, self.light_gx, self.light_gy = f(E, N), f(E, N), f(E, N)
        self.cmd_vx, self.cmd_vy, self.cmd_w = f(E, N), f(E, N), f(E, N)
        self.status = np.zeros(E, np.int32)
        M = cfg.num_objects
        self.ox, self.oy, self.otheta = f(E, M), f(E, M), f(E, M)
        self.ovx, self.ovy, self.ow = f(E, M), f(E, M), f(E, M)
        self.ows_acc = np.full((E, MAX_OBJECTS, OWS_COLS, OWS_WORDS), -1.0, np.float32)
        self.nbr_count = np.zeros((E, N), np.uint32)
        self.sleep_time, self.osleep = f(E, N), f(E, M)       # b2Body::m_sleepTime (< 0: asleep)
        self.bot_mode = np.full((E, N), DRIVE_MOTORS, np.uint8)       # DRIVE_MIXED: per-kilobot drive law
        self._st = State()
        for name, _t in State._fields_:
            arr = getattr(self, name, None)
            if arr is not None:
                setattr(self._st, name, arr.ctypes.data_as(_t))

    def set_poses_m(self, xy_m, theta):
        """Poses in metres / radians (float64 ok), stored as fp32 world units like body.py:32-34."""
        xy = np.asarray(xy_m, np.float64) * WORLD_SCALE
        self.x[...] = xy[..., 0].astype(np.float32)
        self.y[...] = xy[..., 1].astype(np.float32)
        self.theta[...] = np.asarray(theta, np.float32)
        self.ws_cnt[...] = 0
        self.ows_acc[...] = -1.0
        self.sleep_time[...] = 0.0

    def poses_m(self):
        return np.stack([self.x.astype(np.float64) / WORLD_SCALE, self.y.astype(np.float64) / WORLD_SCALE,
                         self.theta.astype(np.float64)], -1)

    def set_actions(self, actions):
        a = None if actions is None else np.ascontiguousarray(actions, np.float32)
        r = lib().kbo_set_actions(C.byref(self.cfg), C.byref(self._st),
                                  None if a is None else a.ctypes.data_as(_PF))
        assert r == 0, r

    def step(self, n_substeps=1, light_action=None, flags=0, threads=1):
        la = None if light_action is None else np.ascontiguousarray(light_action, np.float32)
        r = lib().kbo_step(C.byref(self.cfg), C.byref(self._st),
                           None if la is None else la.ctypes.data_as(_PF), n_substeps, flags, threads)
        assert r == 0, r

    def light_sense(self, light_action=None):
        """The sensing point of one substep on its own (kbo_light_sense): light.step + sensing into light_value / gx / gy."""
        la = None if light_action is None else np.ascontiguousarray(light_action, np.float32)
        r = lib().kbo_light_sense(C.byref(self.cfg), C.byref(self._st), None if la is None else la.ctypes.data_as(_PF))
        assert r == 0, r

    def sense(self, radius_m):
        """Neighbour counts [E, N] (uint32) on the current poses: brute force over all pairs (kbo_sense)."""
        out = np.zeros((self.cfg.num_envs, self.cfg.num_bots), np.uint32)
        r = lib().kbo_sense(C.byref(self.cfg), C.byref(self._st), C.c_float(radius_m), out.ctypes.data_as(_PU32))
        assert r == 0, r
        return out

    def reset(self, seed=0, mean=(0.0, 0.0), std=0.1, random_theta=False, random_velocity=False, resolve=True, env_offset=0):
        """The specification of kb_reset: Philox-keyed Gaussian spawn + (resolve) one world step at rest."""
        rp = ResetParams()
        rp.seed, rp.env_offset = int(seed) & 0xFFFFFFFFFFFFFFFF, int(env_offset)
        rp.mean[0], rp.mean[1], rp.std = float(mean[0]), float(mean[1]), float(std)
        rp.random_theta, rp.random_velocity, rp.resolve = int(bool(random_theta)), int(bool(random_velocity)), int(bool(resolve))
        r = lib().kbo_reset(C.byref(self.cfg), C.byref(self._st), C.byref(rp))
        assert r == 0, r
        if resolve:
            self.step(1, flags=STEP_NO_DRIVE)

    def count_contacts(self, env=0, with_objects=False):
        nb, nw, no = C.c_int32(0), C.c_int32(0), C.c_int32(0)
        lib().kbo_count_contacts(C.byref(self.cfg), C.byref(self._st), env, C.byref(nb), C.byref(nw), C.byref(no))
        return (nb.value, nw.value, no.value) if with_objects else (nb.value, nw.value)

    def set_objects_m(self, xy_m, theta=None):
        """Object poses in metres / radians, at rest (Body.__init__, body.py:32-38)."""
        xy = np.asarray(xy_m, np.float64) * WORLD_SCALE
        self.ox[...] = xy[..., 0].astype(np.float32)
        self.oy[...] = xy[..., 1].astype(np.float32)
        self.otheta[...] = 0.0 if theta is None else np.asarray(theta, np.float32)
        self.ovx[...] = 0
        self.ovy[...] = 0
        self.ow[...] = 0
        self.ows_acc[...] = -1.0
        self.osleep[...] = 0.0

    def objects_m(self):
        return np.stack([self.ox.astype(np.float64) / WORLD_SCALE, self.oy.astype(np.float64) / WORLD_SCALE,
                         self.otheta.astype(np.float64)], -1)


def logf(x):
    return float(lib().kbo_logf(C.c_float(x)))


def philox4x32_10(counter, key):
    c = (C.c_uint32 * 4)(*counter)
    k = (C.c_uint32 * 2)(*key)
    o = (C.c_uint32 * 4)()
    lib().kbo_philox4x32_10(c, k, o)
    return tuple(int(v) for v in o)


def sincosf(x):
    s, c = C.c_float(0), C.c_float(0)
    lib().kbo_sincosf(C.c_float(x), C.byref(s), C.byref(c))
    return s.value, c.value
