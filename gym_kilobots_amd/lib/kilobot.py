"""Kilobot views (reference gym_kilobots/lib/kilobot.py).  Each class fixes the drive law the HIP
kernel applies to every kilobot of the env (KB_DRIVE_*); the per-kilobot methods read and write
the batched device state."""
import numpy as np

from .. import _native as nat
from ..spaces import Box
from .body import Circle, _world_scale  # noqa: F401


class Kilobot(Circle):
    _radius = 0.0165

    _leg_front = np.array([.0, _radius])
    _leg_left = np.array([-0.013, -.009])
    _leg_right = np.array([+0.013, -.009])
    _light_sensor = np.array([.0, -_radius + .001])
    _led = np.array([.011, .01])

    _max_linear_velocity = 0.01  # meters / s
    _max_angular_velocity = 0.5 * np.pi  # radians / s

    _density = 1.0
    _friction = 0.0
    _restitution = 0.0

    _linear_damping = .8
    _angular_damping = .8

    drive_mode = nat.DRIVE_MOTORS

    def __init__(self, world, position=None, orientation=None, light=None):
        super().__init__(world=world, position=position, orientation=orientation, radius=self._radius)
        self._motor_left = 0
        self._motor_right = 0
        self._body_color = (150, 150, 150)
        self._highlight_color = (255, 255, 255)
        self._turn_direction = None
        self._setup()

    def _arrays(self):
        return ('x', 'y', 'theta')

    def light_sensor_pos(self):
        return self.get_world_point((0.0, -self._radius))

    def get_ambientlight(self):
        be = self._world.backend
        if be is None or getattr(be, 'light_value', None) is None:
            return 0
        v = self._get('light_value')
        return v if v else 0

    def set_light_value_and_gradient(self, value, gradient):
        """kilobot.py:50-52: what the reference env calls per kilobot per substep (kilobots_env.py:179-180).  Here the device
        evaluates the light at every sensor itself; a caller that sets a value by hand writes this kilobot's sensing outputs,
        which hold until the next sensing point."""
        be = self._world.backend
        if self._live() and getattr(be, 'light_value', None) is not None:
            self._set('light_value', float(value))
            if gradient is not None:
                self._set('light_gx', float(gradient[0]))
                self._set('light_gy', float(gradient[1]))

    def set_motors(self, left, right):
        self._motor_left = left
        self._motor_right = right
        if self._live() and self._world.backend.motor_l is not None:
            self._set('motor_l', int(left))
            self._set('motor_r', int(right))

    def get_motors(self):
        if self._live() and self._world.backend.motor_l is not None:
            return int(self._get('motor_l')), int(self._get('motor_r'))
        return self._motor_left, self._motor_right

    def switch_directions(self):
        if self._turn_direction == 'left':
            self.turn_right()
        else:
            self.turn_left()

    def turn_right(self):
        self._turn_direction = 'right'
        self.set_motors(0, 255)
        self.set_color((255, 0, 0))

    def turn_left(self):
        self._turn_direction = 'left'
        self.set_motors(255, 0)
        self.set_color((0, 255, 0))

    def set_color(self, color):
        self._highlight_color = color

    def step(self, time_step):
        raise NotImplementedError('kilobots are stepped in bulk by the env (one HIP launch per substep batch)')

    # ---- where does this kilobot's program run? ----------------------------------------------------------------------
    @classmethod
    def _user_code(cls, name):
        """Is method `name` of this class user code (defined outside this module and not an empty body)?"""
        impl = getattr(cls, name)
        owner = next(c for c in cls.__mro__ if name in c.__dict__)
        code = getattr(impl, '__code__', None)
        noop = code is not None and code.co_code == MotorKilobot._loop.__code__.co_code and not code.co_names
        return owner.__module__ != __name__ and not noop

    @classmethod
    def _host_programmed(cls):
        """The reference's extension point is a Kilobot subclass with its own _loop / _setup (kilobot.py:86-88,164-168):
        Python run per kilobot per substep, between the light sensing and the motor law.  Such a class is stepped with its
        _loop on the HOST (KilobotsEnv._step_host_programmed: kb_light_sense -> _loop of every kilobot -> kb_step(1) with
        the motor law and the world step on the device); the library's own classes run entirely on the device."""
        return cls._user_code('_loop')

    def _assert_steppable(self):
        """A subclass that overrides step() itself writes a b2Body's velocities directly (kilobot.py:123-127): there is no
        b2Body here.  Refused loudly when the scene is uploaded instead of being silently ignored."""
        cls = type(self)
        if cls._user_code('step'):
            raise NotImplementedError(
                '%s.step is user code that drives a Box2D body directly; the device step implements the drive laws of Kilobot '
                '(motors), SimplePhototaxisKilobot, SimpleVelocityControlKilobot, SimpleAccelerationControlKilobot and '
                'PhototaxisKilobot.  Program the kilobot through _loop / set_motors (runs on the host, the physics on the device) '
                'or derive from one of those classes without overriding step.' % cls.__name__)
        if cls._host_programmed() and cls.drive_mode not in (nat.DRIVE_MOTORS, nat.DRIVE_PHOTOTAXIS):
            raise NotImplementedError(
                '%s overrides _loop but derives from a class whose step() is not the motor law (%s): a host-side _loop can only '
                'drive the motors (Kilobot.step, kilobot.py:86-127)' % (cls.__name__, cls.__mro__[1].__name__))

    def _host_loop(self):
        """What the host runs for this kilobot at each substep of a host-programmed env: the user's _loop."""
        self._loop()

    @classmethod
    def get_radius(cls):
        return cls._radius

    def _setup(self):
        raise NotImplementedError('Kilobot subclass needs to implement _setup')

    def _loop(self):
        raise NotImplementedError('Kilobot subclass needs to implement _loop')


class MotorKilobot(Kilobot):
    """Base motor law with user-set motor values (reference Kilobot.step, kilobot.py:86-127; the
    reference base class is abstract, this concrete subclass exposes the law itself)."""
    drive_mode = nat.DRIVE_MOTORS

    def _setup(self):
        pass

    def _loop(self):
        pass


class SimplePhototaxisKilobot(Kilobot):
    drive_mode = nat.DRIVE_SIMPLE_PHOTOTAXIS

    def _setup(self):
        self.turn_left()

    def _loop(self):
        pass

    def light_sensor_pos(self):
        return self.get_position()


class SimpleVelocityControlKilobot(Kilobot):
    _density = 2.0
    drive_mode = nat.DRIVE_VELOCITY

    action_space = Box(np.array([.0, -Kilobot._max_angular_velocity]),
                       np.array([Kilobot._max_linear_velocity, Kilobot._max_angular_velocity]),
                       dtype=np.float64)
    state_space = Box(np.array([-np.inf, -np.inf, -np.inf]), np.array([np.inf, np.inf, np.inf]), dtype=np.float64)

    def __init__(self, world, *, velocity=None, **kwargs):
        super().__init__(world=world, light=None, **kwargs)
        if velocity is not None:
            self._velocity = np.asarray(velocity, dtype=np.float64)
        else:
            self._velocity = np.random.rand(2) * np.array([self._max_linear_velocity, 2 * self._max_angular_velocity])
            self._velocity[1] -= self._max_angular_velocity

    def set_action(self, action):
        if action is not None:
            action = np.minimum(action, self.action_space.high)
            action = np.maximum(action, self.action_space.low)
            self._velocity = action
        else:
            self._velocity = np.array([.0, .0])
        if self._live():
            self._set('v', self._velocity[0])
            self._set('w', self._velocity[1])

    def get_action(self):
        if self._live():
            return np.array([self._get('v'), self._get('w')])
        return self._velocity

    def _setup(self):
        pass

    def _loop(self):
        pass

    def set_color(self, color):
        self._body_color = color


class SimpleAccelerationControlKilobot(SimpleVelocityControlKilobot):
    _density = 2.0
    drive_mode = nat.DRIVE_ACCEL

    action_space = Box(np.array([-.005, -.2 * np.pi]), np.array([.005, .2 * np.pi]), dtype=np.float64)
    state_space = Box(np.array([-np.inf, -np.inf, -np.inf, .0, -Kilobot._max_angular_velocity]),
                      np.array([np.inf, np.inf, np.inf, Kilobot._max_linear_velocity, Kilobot._max_angular_velocity]),
                      dtype=np.float64)

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self._acceleration = np.array([.0, .0])

    def get_state(self):
        pose = super().get_state()
        if self._live():
            return pose + (self._get('v'), self._get('w'))
        return pose + tuple(self._velocity)

    def set_action(self, action):
        if action is not None:
            action = np.minimum(action, self.action_space.high)
            action = np.maximum(action, self.action_space.low)
            self._acceleration = action
        else:
            self._acceleration = np.array([.0, .0])
        if self._live():
            self._set('acc_v', self._acceleration[0])
            self._set('acc_w', self._acceleration[1])

    def get_action(self):
        return self._acceleration


class PhototaxisKilobot(Kilobot):
    drive_mode = nat.DRIVE_PHOTOTAXIS

    def __init__(self, world, position=None, orientation=None, light=None):
        super().__init__(world=world, position=position, orientation=orientation, light=light)
        # host twin of the device law's per-kilobot state (kilobot.py:307-313); only used in host-programmed envs
        self._pt_threshold = np.float32(-np.inf)
        self._pt_update_counter = 0
        self._pt_no_change_counter = 0

    def _setup(self):
        self.turn_left()

    def _loop(self):
        pass

    def _host_loop(self):
        """kilobot.py:318-333 in the arithmetic of the device law (fp32 threshold), for envs that mix this class with
        host-programmed kilobots: there every _loop runs on the host."""
        if type(self)._host_programmed():
            return self._loop()
        if self._pt_update_counter % 6:
            self._pt_update_counter += 1
            return
        self._pt_update_counter += 1
        meas = np.float32(self.get_ambientlight())
        if meas > self._pt_threshold or self._pt_no_change_counter >= 15:
            self._pt_threshold = np.float32(meas + np.float32(0.01))
            self.switch_directions()
            self._pt_no_change_counter = 0
        else:
            self._pt_no_change_counter += 1
