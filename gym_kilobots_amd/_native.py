"""ctypes binding of libkilobots_hip.so (include/kilobots_hip.h).

The library is the product: there is no CPU fallback.  If it is missing this module raises
ImportError-like RuntimeError at load() time with the build command.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('KB_HIP_LIB', os.path.join(HERE, 'libkilobots_hip.so'))

KB_OK, KB_EINVAL, KB_ENOTBOUND, KB_EHIP, KB_ELDS = 0, -1, -2, -3, -4
DRIVE_VELOCITY, DRIVE_ACCEL, DRIVE_MOTORS, DRIVE_SIMPLE_PHOTOTAXIS, DRIVE_PHOTOTAXIS, DRIVE_MIXED = range(6)
LIGHT_NONE, LIGHT_CIRCULAR, LIGHT_GRADIENT, LIGHT_MOMENTUM, LIGHT_COMPOSITE = range(5)
MAX_LIGHTS = 4
STEP_NO_DRIVE = 1
MAX_OBJECTS = 8
MAX_POLY_VERTS = 4
SHAPE_CIRCLE, SHAPE_BOX, SHAPE_POLYGON = range(3)
OWS_COLS, OWS_WORDS = 12, 6
MAX_BOTS = 1024
DAMPING_PADE, DAMPING_LINEAR = 0, 1
WORLD_SCALE = 25.0    # reference gym_kilobots/lib/body.py:7


class KbConfig(C.Structure):
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


class KbResetParams(C.Structure):
    _fields_ = [('seed', C.c_uint64), ('env_offset', C.c_int32), ('mean', C.c_float * 2), ('std', C.c_float),
                ('random_theta', C.c_int32), ('random_velocity', C.c_int32), ('resolve', C.c_int32)]


_P = C.c_void_p

BUFFER_FIELDS = ['x', 'y', 'theta', 'v', 'w', 'acc_v', 'acc_w', 'motor_l', 'motor_r',
                 'pt_threshold', 'pt_update', 'pt_nochange', 'pt_dir',
                 'light_x', 'light_y', 'light_vx', 'light_vy',
                 'ox', 'oy', 'otheta', 'ovx', 'ovy', 'ow',
                 'ws_key', 'ws_acc', 'ws_cnt',
                 'light_value', 'light_gx', 'light_gy', 'cmd_vx', 'cmd_vy', 'cmd_w', 'status', 'scratch', 'ows_acc', 'nbr_count',
                 'sleep_time', 'osleep', 'bot_mode']


class KbBuffers(C.Structure):
    _fields_ = [(n, _P) for n in BUFFER_FIELDS]


EXPORTS = ['kb_create', 'kb_destroy', 'kb_bind', 'kb_set_actions', 'kb_step', 'kb_get_poses', 'kb_get_state', 'kb_sense', 'kb_light_sense', 'kb_reset',
           'kb_lds_bytes', 'kb_resident_envs_per_cu', 'kb_contact_capacity', 'kb_lds_staging_entries', 'kb_scratch_bytes', 'kb_light_action_dim', 'kb_light_count', 'kb_block_threads', 'kb_set_block_threads',
           'kb_last_error', 'kb_version']

_lib = None


class KilobotsHipError(RuntimeError):
    pass


STATUS_BITS = {
    1: 'contact capacity overflow: contacts were dropped (raise kb_config.contact_capacity or spread the spawn)',
    2: 'warm-start slot overflow: a kilobot touches more partners than kb_config.ws_slots, their impulses are not carried over',
    4: 'a device staging limit was hit (more than 64 kilobots on one fixture, 255 kilobot-object contacts in one env or 63 partners in one cell pair)',
    8: 'continuous step skipped for some kilobots: more kilobots near the walls in one substep than the staging area holds',
}


class KilobotsStatusError(KilobotsHipError):
    """A capacity limit of the device step was hit: the trajectories of the flagged envs are degraded."""


def describe_status(bits):
    return '; '.join(msg for b, msg in sorted(STATUS_BITS.items()) if bits & b) or 'ok'


def load():
    """dlopen the HIP library; fails loudly when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise KilobotsHipError(
            'libkilobots_hip.so is missing (%s). Build it with `python -m gym_kilobots_amd.build` '
            '(needs hipcc; cross-compiles for gfx950 without a GPU). There is no CPU fallback.' % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    lib.kb_create.argtypes = [C.POINTER(KbConfig), C.POINTER(_P)]
    lib.kb_create.restype = C.c_int
    lib.kb_destroy.argtypes = [_P]
    lib.kb_destroy.restype = None
    lib.kb_bind.argtypes = [_P, C.POINTER(KbBuffers)]
    lib.kb_bind.restype = C.c_int
    lib.kb_set_actions.argtypes = [_P, _P, _P]
    lib.kb_set_actions.restype = C.c_int
    lib.kb_step.argtypes = [_P, _P, _P, C.c_int, C.c_int, _P]
    lib.kb_step.restype = C.c_int
    lib.kb_get_poses.argtypes = [_P, _P, _P]
    lib.kb_get_poses.restype = C.c_int
    lib.kb_get_state.argtypes = [_P, _P, _P]
    lib.kb_get_state.restype = C.c_int
    lib.kb_sense.argtypes = [_P, C.c_float, _P, _P]
    lib.kb_sense.restype = C.c_int
    lib.kb_light_sense.argtypes = [_P, _P, _P]
    lib.kb_light_sense.restype = C.c_int
    lib.kb_reset.argtypes = [_P, C.POINTER(KbResetParams), _P]
    lib.kb_reset.restype = C.c_int
    for name in ('kb_lds_bytes', 'kb_resident_envs_per_cu', 'kb_contact_capacity', 'kb_lds_staging_entries', 'kb_block_threads', 'kb_light_action_dim', 'kb_light_count'):
        getattr(lib, name).argtypes = [_P]
        getattr(lib, name).restype = C.c_int
    lib.kb_scratch_bytes.argtypes = [_P]
    lib.kb_scratch_bytes.restype = C.c_size_t
    lib.kb_set_block_threads.argtypes = [_P, C.c_int]
    lib.kb_set_block_threads.restype = C.c_int
    lib.kb_last_error.argtypes = []
    lib.kb_last_error.restype = C.c_char_p
    lib.kb_version.argtypes = []
    lib.kb_version.restype = C.c_char_p
    _lib = lib
    return lib


def check(rc, what):
    if rc != KB_OK:
        msg = load().kb_last_error().decode('utf-8', 'replace')
        raise KilobotsHipError('%s failed (%d): %s' % (what, rc, msg))


def default_config(num_envs, num_bots, drive_mode=DRIVE_VELOCITY, light_type=LIGHT_NONE, **kw):
    """Reference defaults: kilobots_env.py:19,25-28; kilobot.py:9,25-30,214; body.py:11-16; light.py:49-54,152."""
    inf = float('inf')
    c = KbConfig()
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
    c.light_lo[0] = c.light_lo[1] = -inf
    c.light_hi[0] = c.light_hi[1] = inf
    c.light_act_lo[0] = c.light_act_lo[1] = -0.01
    c.light_act_hi[0] = c.light_act_hi[1] = 0.01
    c.light_max_velocity = inf
    c.ws_slots = 32     # contacts per kilobot whose impulse is carried over (Box2D keeps every b2Contact; 32 covers a dense overlapping spawn)
    c.obj_density, c.obj_friction = 2.0, 0.01
    c.obj_linear_damping = c.obj_angular_damping = 0.8
    c.toi_walls = 1      # b2World continuousPhysics defaults to true
    c.wall_friction = 0.2  # b2FixtureDef default (the arena chain, kilobots_env.py:46-51)
    c.solver_mode = 0
    c.damping_model = DAMPING_PADE
    c.sense_radius = 0.0
    c.contact_capacity = 0
    c.allow_sleep = 1      # b2World(gravity=(0, 0), doSleep=True) of kilobots_env.py:45; 0 drops the sleep state
    c.light_count = 1
    for i in range(MAX_LIGHTS):
        c.light_kind[i] = LIGHT_CIRCULAR
        c.lightc_radius[i] = 0.2
        c.lightc_max_velocity[i] = inf
        for k in range(2):
            c.lightc_lo[i][k], c.lightc_hi[i][k] = -inf, inf
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
