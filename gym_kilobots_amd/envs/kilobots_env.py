"""KilobotsEnv: the reference's gym.Env surface (gym_kilobots/envs/kilobots_env.py) on top of the
batched HIP world step.

Subclasses configure a scene exactly as with the reference: `_configure_environment` creates
kilobots / a light through the same constructors, passing `self.world`:

    class MyEnv(KilobotsEnv):
        def _configure_environment(self):
            self._light = CircularGradientLight(position=np.zeros(2))
            for p in positions:
                self._add_kilobot(PhototaxisKilobot(self.world, position=p, light=self._light))

`reset()` then uploads the scene to the GPU and `step(action)` runs the 10-substep loop of
kilobots_env.py:168-190 as ONE kernel launch.  `num_envs > 1` replicates the configured scene into
independent worlds (see BatchedKilobotsEnv for the tensor-level API)."""
import abc

import numpy as np
import torch

from .. import _native as nat
from ..lib.body import World, Body, _world_scale  # noqa: F401
from ..lib.kilobot import Kilobot
from ..lib.light import Light, CircularGradientLight, GradientLight, MomentumLight, CompositeLight


class UnknownObjectException(Exception):
    pass


class UnknownLightTypeException(Exception):
    pass


def _default_sim_factory(*args, **kwargs):
    from ..sim import KilobotSim
    return KilobotSim(*args, **kwargs)


class KilobotsEnv(object):
    metadata = {'render.modes': ['human']}

    world_size = world_width, world_height = 2., 1.5
    screen_size = screen_width, screen_height = 1200, 900

    _observe_objects = False
    _observe_light = True

    __sim_steps_per_second = 10
    __sim_velocity_iterations = 10
    __sim_position_iterations = 10
    __steps_per_action = 10

    def __new__(cls, **kwargs):
        cls.sim_steps_per_second = cls.__sim_steps_per_second
        cls.sim_step = 1. / cls.__sim_steps_per_second
        cls.world_x_range = -cls.world_width / 2, cls.world_width / 2
        cls.world_y_range = -cls.world_height / 2, cls.world_height / 2
        cls.world_bounds = (np.array([-cls.world_width / 2, -cls.world_height / 2]),
                            np.array([cls.world_width / 2, cls.world_height / 2]))
        return super(KilobotsEnv, cls).__new__(cls)

    def __init__(self, num_envs=1, device=None, sim_factory=None, on_status='raise', allow_sleep=True, **kwargs):
        # on_status: what reset() / step() do when the device step reports a capacity overflow (dropped contacts, lost
        # warm-start impulses, staging limits: include/kilobots_hip.h, kb_buffers.status): 'raise' | 'warn' | 'ignore'
        if on_status not in ('raise', 'warn', 'ignore'):
            raise ValueError("on_status must be 'raise', 'warn' or 'ignore'")
        self._on_status = on_status
        # allow_sleep: the reference creates its world with doSleep=True (kilobots_env.py:45): islands at rest fall asleep
        self._allow_sleep = bool(allow_sleep)
        self.__sim_steps = 0
        self.__reset_counter = 0
        self.__seed = 0
        self.num_envs = int(num_envs)
        self._device = device
        self._sim_factory = sim_factory or _default_sim_factory
        self._sim = None
        self._sim_signature = None
        self._host_programmed = False

        # the "world": records bodies created by _configure_environment (kilobots_env.py:45-51)
        self.world = World()
        self._real_time = False
        self._kilobots = []
        self._objects = []
        self._light = None
        self._state_cache = None     # (world.version, state dict) of the last device read
        self._status_seen = None     # (world.version, status words) of the last device read
        self._screen = None
        self.render_mode = 'human'
        self.video_path = None

        # like the reference constructor: configure once, then drop the kilobots again
        # (kilobots_env.py:67-68) -- an env has no kilobots until the first reset()
        self._configure_environment()
        self._kilobots = []
        self.world.clear()

    # ------------------------------------------------------------------ reference properties
    @property
    def _sim_steps(self):
        return self.__sim_steps

    @property
    def kilobots(self):
        return tuple(self._kilobots)

    @property
    def num_kilobots(self):
        return len(self._kilobots)

    @property
    def objects(self):
        return tuple(self._objects)

    @property
    def action_space(self):
        if self._light:
            return self._light.action_space

    @property
    def observation_space(self):
        return NotImplemented

    @property
    def state_space(self):
        return NotImplemented

    @property
    def _steps_per_action(self):
        return self.__steps_per_action

    @property
    def sim(self):
        """The device-side simulator (gym_kilobots_amd.sim.KilobotSim); None before the first reset()."""
        return self._sim

    def _add_kilobot(self, kilobot):
        self._kilobots.append(kilobot)

    def _add_object(self, body):
        self._objects.append(body)

    @abc.abstractmethod
    def _configure_environment(self):
        raise NotImplementedError

    # ------------------------------------------------------------------ state
    def get_state(self):
        if self._sim is None:
            return {'kilobots': np.array([k.get_state() for k in self._kilobots]),
                    'objects': np.array([o.get_state() for o in self._objects]),
                    'light': self._light.get_state() if self._light else None}
        # env.step asks for the state three times (kilobots_env.py:166,201,204): the device is read once per change.
        # (Writes through the body / light views and the env's own launches bump world.version; code that writes the
        # sim's tensors directly calls env.world.touch().)
        if self._state_cache is not None and self._state_cache[0] == self.world.version:
            return {k: (None if v is None else v.copy()) for k, v in self._state_cache[1].items()}
        state = self._read_state()
        self._state_cache = (self.world.version, state)
        return {k: (None if v is None else v.copy()) for k, v in state.items()}

    def _read_state(self):
        # one kernel + ONE copy instead of 3 (N + M) SWIG reads: kilobots, objects and the status word (kb_get_state)
        poses, oposes, status = self._sim.host_state()
        self._status_seen = (self.world.version, status)
        kb = (poses[0] if self.num_envs == 1 else poses).astype(np.float64)
        if self._sim.drive_mode == nat.DRIVE_ACCEL:
            vw = torch.stack([self._sim.v, self._sim.w], -1).cpu().numpy().astype(np.float64)
            kb = np.concatenate([kb, vw[0] if self.num_envs == 1 else vw], -1)
        light = None
        if self._light is not None:
            if self.num_envs == 1:
                light = np.asarray(self._light.get_state(), dtype=np.float64)
            elif isinstance(self._light, GradientLight):
                light = self._sim.light_x.double().cpu().numpy()[:, None]
            else:
                # per env the concatenation of every component's get_state(): (x, y), or (x, y, vx, vy) for a MomentumLight
                # (light.py:256-257, 318-319) -- the same layout as the single-env path
                comps = self._light.lights if isinstance(self._light, CompositeLight) else (self._light,)
                cols = []
                for i, c in enumerate(comps):
                    names = ('light_x', 'light_y') + (('light_vx', 'light_vy') if isinstance(c, MomentumLight) else ())
                    cols += [getattr(self._sim, n).reshape(self.num_envs, -1)[:, i] for n in names]
                light = torch.stack(cols, -1).double().cpu().numpy()
        objs = np.array([])
        if self._objects:
            objs = (oposes[0] if self.num_envs == 1 else oposes).astype(np.float64)
        return {'kilobots': kb, 'objects': objs, 'light': light}

    def get_observation(self):
        return self.get_state()

    @abc.abstractmethod
    def get_reward(self, state, action, new_state):
        raise NotImplementedError

    def has_finished(self, state, action):
        return False

    def get_info(self, state, action):
        return ""

    def destroy(self):
        del self._objects[:]
        del self._kilobots[:]
        self._light = None
        self.world.clear()
        self.world.backend = None
        self._screen = None

    def close(self):
        self.destroy()
        if self._sim is not None:
            self._sim.close()
            self._sim = None
            self._sim_signature = None

    def seed(self, seed=None):
        if seed is not None:
            self.__seed = seed
        return [self.__seed]

    # ------------------------------------------------------------------ scene upload
    def _upload_scene(self):
        kbs = self._kilobots
        if len(kbs) == 0:
            raise ValueError('the configured scene has no kilobots')
        specs, fixture_body = [], []
        for i, ob in enumerate(self._objects):
            if isinstance(ob, Kilobot) or not hasattr(ob, '_shape_spec'):
                raise UnknownObjectException('pushable objects must be lib.Circle, lib.Quad / CornerQuad or a lib.Polygon')
            try:
                fixtures = ob._shape_spec()
            except NotImplementedError as err:
                raise UnknownObjectException(str(err))
            specs.extend(fixtures)
            fixture_body.extend([i] * len(fixtures))
        if len(specs) > nat.MAX_OBJECTS:
            raise UnknownObjectException('at most %d fixtures (convex parts of all objects) per env' % nat.MAX_OBJECTS)
        for k in kbs:
            if not isinstance(k, Kilobot):
                raise TypeError('kilobots must derive from gym_kilobots_amd.lib.Kilobot')
            k._assert_steppable()
        kinds = {type(k).drive_mode for k in kbs}
        motor_family = kinds <= {nat.DRIVE_MOTORS, nat.DRIVE_PHOTOTAXIS}
        # Kilobots with their own _loop (the reference's extension point, kilobot.py:86-88,164-168) and mixes of the classes
        # that share Kilobot.step (the motor law) are "host-programmed": every substep the device senses the light, the
        # host runs each kilobot's _loop, and the device applies the motor law and steps the world (_step_host_programmed).
        self._host_programmed = any(type(k)._host_programmed() for k in kbs)
        if self._host_programmed:
            if not motor_family:
                raise ValueError('host-programmed kilobots (own _loop) only mix with classes that use the motor law of '
                                 'Kilobot.step (got drive laws %s)' % sorted(kinds))
            if self.num_envs != 1:
                raise ValueError('kilobots with their own _loop are Python objects of ONE world: use num_envs=1 '
                                 '(the batched device laws are the classes of gym_kilobots_amd.lib.kilobot)')
            kinds = {nat.DRIVE_MOTORS}
        # Kilobots of several classes in one env (the reference steps whatever is in _kilobots, kilobots_env.py:183-184): the
        # device step takes the drive law per kilobot (KB_DRIVE_MIXED) and the fixture density per law (the velocity /
        # acceleration kilobots are twice as dense, kilobot.py:214)
        mixed = len(kinds) > 1
        mode_density = [0.0] * 5
        if mixed:
            for law in kinds:
                dens = {float(type(k)._density) for k in kbs if type(k).drive_mode == law}
                if len(dens) != 1:
                    raise ValueError('kilobots of one drive law must share their density (got %s)' % sorted(dens))
                mode_density[law] = dens.pop()
        props = {((0.0 if mixed else float(type(k)._density)), float(type(k)._radius), float(type(k)._linear_damping),
                  float(type(k)._angular_damping)) for k in kbs}
        if len(props) != 1:
            raise ValueError('all kilobots of an env must share radius and damping, and (unless drive laws are mixed) density (got %s)' % sorted(props))
        mode = nat.DRIVE_MIXED if mixed else kinds.pop()
        light_type = nat.LIGHT_NONE
        overrides = dict(world_width=self.world_width, world_height=self.world_height, dt=self.sim_step,
                         vel_iters=self.__sim_velocity_iterations, pos_iters=self.__sim_position_iterations,
                         bot_density=float(type(kbs[0])._density), bot_radius=float(type(kbs[0])._radius),
                         bot_linear_damping=float(type(kbs[0])._linear_damping),
                         bot_angular_damping=float(type(kbs[0])._angular_damping),
                         allow_sleep=1 if self._allow_sleep else 0)      # b2World(gravity=(0, 0), doSleep=True), kilobots_env.py:45
        if mixed:
            overrides['mode_density'] = mode_density
        if self._objects:
            ob0 = type(self._objects[0])
            pad = nat.MAX_OBJECTS - len(specs)
            radii = [sp[1] for sp in specs] + [0.075] * pad
            verts = [[list(v) for v in sp[2]] + [[0.0, 0.0]] * (nat.MAX_POLY_VERTS - len(sp[2])) for sp in specs]
            overrides.update(num_objects=len(self._objects), obj_radius=radii,
                             num_fixtures=len(specs), obj_fixture_body=fixture_body + [0] * pad,
                             obj_shape=[sp[0] for sp in specs] + [0] * pad,
                             obj_nverts=[len(sp[2]) for sp in specs] + [0] * pad,
                             obj_verts=verts + [[[0.0, 0.0]] * nat.MAX_POLY_VERTS] * pad,
                             obj_density=float(ob0._density),
                             obj_friction=float(ob0._friction), obj_linear_damping=float(ob0._linear_damping),
                             obj_angular_damping=float(ob0._angular_damping))
        if self._light is not None:
            lt = self._light
            if isinstance(lt, CompositeLight):
                comps = lt.lights
                if not (1 <= len(comps) <= nat.MAX_LIGHTS) or not all(isinstance(c, CircularGradientLight) for c in comps):
                    raise UnknownLightTypeException('a CompositeLight on the device holds 1..4 circular / momentum lights')
                pad = nat.MAX_LIGHTS - len(comps)
                light_type = nat.LIGHT_COMPOSITE
                overrides.update(
                    light_count=len(comps),
                    light_kind=[nat.LIGHT_MOMENTUM if isinstance(c, MomentumLight) else nat.LIGHT_CIRCULAR for c in comps] + [nat.LIGHT_CIRCULAR] * pad,
                    lightc_radius=[float(c._radius) for c in comps] + [0.2] * pad,
                    lightc_max_velocity=[float(getattr(c, 'max_velocity', np.inf)) for c in comps] + [np.inf] * pad,
                    lightc_lo=[[float(v) for v in c._bounds[0]] for c in comps] + [[0.0, 0.0]] * pad,
                    lightc_hi=[[float(v) for v in c._bounds[1]] for c in comps] + [[0.0, 0.0]] * pad,
                    lightc_act_lo=[[float(v) for v in c._action_bounds[0]] for c in comps] + [[0.0, 0.0]] * pad,
                    lightc_act_hi=[[float(v) for v in c._action_bounds[1]] for c in comps] + [[0.0, 0.0]] * pad)
            elif isinstance(lt, GradientLight):
                light_type = nat.LIGHT_GRADIENT
            elif isinstance(lt, CircularGradientLight):
                light_type = nat.LIGHT_MOMENTUM if isinstance(lt, MomentumLight) else nat.LIGHT_CIRCULAR
                overrides.update(light_radius=float(lt._radius),
                                 light_lo=[float(v) for v in lt._bounds[0]], light_hi=[float(v) for v in lt._bounds[1]],
                                 light_act_lo=[float(v) for v in lt._action_bounds[0]],
                                 light_act_hi=[float(v) for v in lt._action_bounds[1]],
                                 light_max_velocity=float(getattr(lt, 'max_velocity', np.inf)))
            else:
                raise UnknownLightTypeException('light model %s does not run on the device' % type(lt).__name__)
        N = len(kbs)
        sig = (self.num_envs, N, mode, light_type, tuple(sorted((k, str(v)) for k, v in overrides.items())))
        if self._sim is None or sig != self._sim_signature:
            if self._sim is not None:
                self._sim.close()
            kw = dict(overrides)
            if self._device is not None:
                kw['device'] = self._device
            self._sim = self._sim_factory(self.num_envs, N, mode, light_type, debug_outputs=True, **kw)
            self._sim_signature = sig
        sim = self._sim
        poses = np.array([k._init_pose for k in kbs], dtype=np.float64)
        xy = np.broadcast_to(poses[None, :, :2], (self.num_envs, N, 2))
        th = np.broadcast_to(poses[None, :, 2], (self.num_envs, N))
        sim.set_poses_m(xy, th)
        sim.status.zero_()
        if self._objects:
            op = np.array([ob._init_pose for ob in self._objects], dtype=np.float64)
            sim.set_objects_m(np.broadcast_to(op[None, :, :2], (self.num_envs, len(self._objects), 2)),
                              np.broadcast_to(op[None, :, 2], (self.num_envs, len(self._objects))))
        if mode == nat.DRIVE_MIXED:
            laws = np.array([type(k).drive_mode for k in kbs], dtype=np.uint8)
            sim.bot_mode.copy_(torch.from_numpy(np.broadcast_to(laws[None], (self.num_envs, N)).copy()))
        if mode in (nat.DRIVE_VELOCITY, nat.DRIVE_ACCEL, nat.DRIVE_MIXED):
            v0 = np.array([getattr(k, '_velocity', (0.0, 0.0)) for k in kbs], dtype=np.float32)
            sim.v.copy_(torch.from_numpy(np.broadcast_to(v0[None, :, 0], (self.num_envs, N)).copy()))
            sim.w.copy_(torch.from_numpy(np.broadcast_to(v0[None, :, 1], (self.num_envs, N)).copy()))
        if mode in (nat.DRIVE_ACCEL, nat.DRIVE_MIXED):
            sim.acc_v.zero_()
            sim.acc_w.zero_()
        if mode in (nat.DRIVE_MOTORS, nat.DRIVE_PHOTOTAXIS, nat.DRIVE_MIXED):
            ml = np.array([k._motor_left for k in kbs], dtype=np.uint8)
            mr = np.array([k._motor_right for k in kbs], dtype=np.uint8)
            sim.motor_l.copy_(torch.from_numpy(np.broadcast_to(ml[None], (self.num_envs, N)).copy()))
            sim.motor_r.copy_(torch.from_numpy(np.broadcast_to(mr[None], (self.num_envs, N)).copy()))
        if mode in (nat.DRIVE_PHOTOTAXIS, nat.DRIVE_MIXED):
            sim.pt_threshold.fill_(float('-inf'))
            sim.pt_update.zero_()
            sim.pt_nochange.zero_()
            sim.pt_dir.zero_()
        if self._light is not None:
            lt = self._light
            comps = lt.lights if isinstance(lt, CompositeLight) else (lt,)
            for i, c in enumerate(comps):
                c._world = self.world
                c._slot = i if isinstance(lt, CompositeLight) else None
            if isinstance(lt, GradientLight):
                sim.light_x.fill_(float(lt._gradient_angle[0]))
            else:
                lx = torch.tensor([[float(c._position[0]) for c in comps]], dtype=torch.float32).expand(self.num_envs, -1)
                ly = torch.tensor([[float(c._position[1]) for c in comps]], dtype=torch.float32).expand(self.num_envs, -1)
                lvx = torch.tensor([[float(getattr(c, '_velocity', (0.0, 0.0))[0]) for c in comps]], dtype=torch.float32).expand(self.num_envs, -1)
                lvy = torch.tensor([[float(getattr(c, '_velocity', (0.0, 0.0))[1]) for c in comps]], dtype=torch.float32).expand(self.num_envs, -1)
                for name, val in (('light_x', lx), ('light_y', ly), ('light_vx', lvx), ('light_vy', lvy)):
                    dst = getattr(sim, name)
                    dst.copy_(val.reshape(dst.shape).to(dst.device))
            lt._world = self.world
        self.world.backend = sim
        self.world.touch()
        self.world.env_index = 0

    def reset(self):
        self.__reset_counter += 1
        self.destroy()
        self._configure_environment()
        self.__sim_steps = 0
        self._upload_scene()
        # step to resolve (kilobots_env.py:156-157): one world.Step with the bodies at rest
        self._step_world()
        self._check_status('reset()')
        return self.get_observation()

    def _check_status(self, where):
        """Capacity overflows of the device step are never silent: raise / warn with the decoded bits."""
        if self._on_status == 'ignore' or self._sim is None:
            return
        if self._status_seen is not None and self._status_seen[0] == self.world.version:       # came with the state read
            bits = int(np.bitwise_or.reduce(self._status_seen[1], initial=0)) & sum(nat.STATUS_BITS)
        else:
            bits = self._sim.status_bits()
        if not bits:
            return
        self._status_seen = None
        msg = '%s: device step status 0x%x: %s' % (where, bits, nat.describe_status(bits))
        if self._on_status == 'raise':
            raise nat.KilobotsStatusError(msg)
        import warnings
        warnings.warn(msg, RuntimeWarning, stacklevel=3)
        self._sim.status.zero_()          # warn once per occurrence

    def _light_action_tensor(self, action):
        adim = self._light.action_space.shape[0]
        if torch.is_tensor(action):
            a = action.to(device=self._sim.x.device, dtype=torch.float32).reshape(-1, adim)
        else:
            a = torch.as_tensor(np.asarray(action, dtype=np.float32).reshape(-1, adim), device=self._sim.x.device)
        if a.shape[0] == 1 and self.num_envs > 1:
            a = a.expand(self.num_envs, adim)
        return a.contiguous()

    def step(self, action):
        if self._sim is None:
            raise RuntimeError('call reset() before step()')
        state = self.get_state()
        la = None
        if action is not None and self._light:
            la = self._light_action_tensor(action)
        if self._host_programmed:
            self._step_host_programmed(la)
        else:
            # the whole `for i in range(steps_per_action)` loop (kilobots_env.py:168-190) is one launch
            self._sim.step(self.__steps_per_action, light_action=la)
        self.world.touch()
        self.__sim_steps += self.__steps_per_action
        next_state = self.get_state()
        self._check_status('step()')
        observation = self.get_observation()
        reward = self.get_reward(state, action, next_state)
        done = self.has_finished(next_state, action)
        info = self.get_info(next_state, action)
        return observation, reward, done, info

    def _step_world(self):
        self._sim.step(1, flags=nat.STEP_NO_DRIVE)
        self.world.touch()

    # ------------------------------------------------------------------ host-programmed kilobots
    _HOST_ARRAYS = ('motor_l', 'motor_r', 'x', 'y', 'theta')

    def _step_host_programmed(self, la):
        """The substep loop of kilobots_env.py:168-190 for kilobots whose _loop is Python: per substep the device steps the
        light and evaluates it at every sensor (kb_light_sense, :171-180), the host runs every kilobot's _loop on copies of
        the state (get_ambientlight / set_motors / the body getters; kilobot.py:88), and the device applies the motor law
        and world.Step (kb_step(1); :183-188).  Physics and sensing stay on the GPU; only the user's code runs here."""
        sim, world = self._sim, self.world
        names = self._HOST_ARRAYS + (('light_value', 'light_gx', 'light_gy') if self._light is not None else ())
        for _ in range(self.__steps_per_action):
            if self._light is not None:
                sim.light_sense(la)
            cache = {n: getattr(sim, n)[0].cpu().numpy().copy() for n in names}
            world.host_cache, world.host_dirty = cache, set()
            try:
                for k in self._kilobots:
                    k._host_loop()
            finally:
                world.host_cache = None
            for n in sorted(world.host_dirty):
                dst = getattr(sim, n)
                dst[0].copy_(torch.from_numpy(cache[n]).to(dst.device))
            if world.host_dirty & {'x', 'y'}:
                sim.forget_contacts()
            world.host_dirty = set()
            sim.step(1)

    def render(self, mode=None):
        raise NotImplementedError('rendering (pygame viewer) is outside the accelerated hot path; '
                                  'see SURVEY.md section 2, component 8')

    def get_objects(self):
        return self._objects

    def get_kilobots(self):
        return self._kilobots

    def get_light(self):
        return self._light

    def _draw_on_table(self, screen):
        pass

    def _draw_on_top(self, screen):
        pass
