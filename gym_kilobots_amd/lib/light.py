"""Light models (reference gym_kilobots/lib/light.py).  The device kernel evaluates the light at
every kilobot each substep; these host classes carry the parameters, the action / observation
spaces and a NumPy evaluation with the reference's formulas for user code that queries a light."""
import numpy as np

from .. import _native as nat
from ..spaces import Box


class Light(object):
    relative_actions = True
    interpolate_actions = True
    light_type = nat.LIGHT_NONE

    def __init__(self, **kwargs):
        self.observation_space = None
        self.action_space = None
        self._world = None      # set by the env: position then lives in the backend's light_x / light_y

    def step(self, action, time_step):
        raise NotImplementedError

    def get_value(self, position):
        raise NotImplementedError

    def get_gradient(self, position):
        raise NotImplementedError

    def value_and_gradients(self, position):
        return self.get_value(position), self.get_gradient(position)

    def get_state(self):
        raise NotImplementedError


class SinglePositionLight(Light):
    def __init__(self, *, position=None, bounds=None, action_bounds=None, relative_actions=True, **kwargs):
        super().__init__(**kwargs)
        self._position = np.array((.0, .0)) if position is None else np.asarray(position, dtype=np.float64)
        self._bounds = bounds
        if self._bounds is None:
            self._bounds = np.array([-np.inf, -np.inf]), np.array([np.inf, np.inf])
        self._relative_actions = relative_actions
        if not relative_actions:
            raise NotImplementedError('absolute light actions are not on the accelerated path')
        self._action_bounds = action_bounds
        if self._action_bounds is None:
            self._action_bounds = np.array([-0.01, -0.01]), np.array([.01, .01])
        self.action_space = Box(*self._action_bounds, dtype=np.float64)
        self.observation_space = Box(*self._bounds, dtype=np.float64)

    _slot = None     # component index inside a CompositeLight (set by the env)

    def _live(self):
        return self._world is not None and self._world.backend is not None

    def _cell(self, name):
        be, e = self._world.backend, self._world.env_index
        t = getattr(be, name)
        return t[e] if t.dim() == 1 else t[e, self._slot or 0]

    def _put(self, name, value):
        be, e = self._world.backend, self._world.env_index
        t = getattr(be, name)
        if t.dim() == 1:
            t[e] = float(value)
        else:
            t[e, self._slot or 0] = float(value)
        self._world.touch()

    def get_position(self):
        if self._live():
            return np.array([float(self._cell('light_x').item()), float(self._cell('light_y').item())])
        return self._position

    def set_position(self, position):
        self._position = np.asarray(position, dtype=np.float64)
        if self._live():
            self._put('light_x', position[0])
            self._put('light_y', position[1])

    def get_state(self):
        return self.get_position()

    def step(self, action, time_step):
        """Host restatement of light.py:59-75 (the env steps the light on the device instead)."""
        if action is None:
            return
        action = np.asarray(action).squeeze()
        action = np.minimum(np.maximum(action, self._action_bounds[0]), self._action_bounds[1])
        pos = self.get_position() + action * time_step
        self.set_position(np.minimum(np.maximum(pos, self._bounds[0]), self._bounds[1]))

    def get_value(self, position):
        return -1 * np.linalg.norm(position - self.get_position(), axis=1)

    def value_and_gradients(self, position):
        gradients = -1 * (position - self.get_position())
        norms = np.linalg.norm(gradients, axis=1)
        return -1 * norms, gradients / norms[:, None]


class CircularGradientLight(SinglePositionLight):
    light_type = nat.LIGHT_CIRCULAR

    def __init__(self, radius=.2, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self._radius = radius

    def get_value(self, position):
        return self.value_and_gradients(position)[0]

    def get_gradient(self, position):
        return self.value_and_gradients(position)[1]

    def value_and_gradients(self, position):
        gradient = -1 * (np.asarray(position, dtype=np.float64) - self.get_position())
        norm_gradient = np.linalg.norm(gradient, axis=1)
        value = np.ones(gradient.shape[0])
        value -= norm_gradient / self._radius
        value = np.maximum(np.minimum(value, 1.), .0)
        value *= 255
        with np.errstate(invalid='ignore', divide='ignore'):
            gradient /= norm_gradient[:, None]
        gradient[norm_gradient > self._radius] *= .0
        return value, gradient


class GradientLight(Light):
    """light.py:218-271.  The reference's value/gradient only broadcast for exactly two kilobots; the device
    evaluates the evident intent: value = position . gradient_vec, gradient = gradient_vec for every kilobot."""
    relative_actions = False
    interpolate_actions = False
    light_type = nat.LIGHT_GRADIENT

    def __init__(self, angle=.0):
        super().__init__()
        self._gradient_angle = np.array([float(np.asarray(angle).ravel()[0])])
        self._bounds = np.array([-np.pi]), np.array([np.pi])
        self._action_bounds = 2 * np.array([-np.pi]), 2 * np.array([np.pi])
        self.observation_space = Box(*self._bounds, dtype=np.float64)
        self.action_space = Box(*self._action_bounds, dtype=np.float64)

    def _live(self):
        return self._world is not None and self._world.backend is not None

    def get_state(self):
        if self._live():
            return np.array([float(self._world.backend.light_x[self._world.env_index].item())])
        return self._gradient_angle

    def set_angle(self, angle):
        self._gradient_angle = np.array([float(angle)])
        if self._live():
            self._world.backend.light_x[self._world.env_index] = float(angle)
            self._world.touch()

    @property
    def _gradient_vec(self):
        a = float(self.get_state()[0])
        return np.r_[np.cos(a), np.sin(a)]

    def step(self, action, time_step):
        if action is None:
            return
        a = float(np.minimum(np.maximum(np.asarray(action).ravel()[0], self._action_bounds[0][0]), self._action_bounds[1][0]))
        if a < self._bounds[0][0]:
            a += 2 * np.pi
        if a > self._bounds[1][0]:
            a -= 2 * np.pi
        self.set_angle(a)

    def get_value(self, position):
        return np.asarray(position, dtype=np.float64).dot(self._gradient_vec)

    def get_gradient(self, position):
        return self._gradient_vec


class MomentumLight(CircularGradientLight):
    """light.py:274-319: the action accelerates the light; speed capped at max_velocity."""
    interpolate_actions = False
    light_type = nat.LIGHT_MOMENTUM

    def __init__(self, velocity=None, max_velocity=None, action_bounds=None, **kwargs):
        super().__init__(action_bounds=action_bounds, **kwargs)
        self._velocity = np.array([.0, .0]) if velocity is None else np.asarray(velocity, dtype=np.float64)
        self.max_velocity = np.inf if max_velocity is None else max_velocity
        mv = self.max_velocity
        self._obs_bounds = np.r_[self._bounds[0], [-mv, -mv]], np.r_[self._bounds[1], [mv, mv]]
        self.observation_space = Box(*self._obs_bounds, dtype=np.float64)

    def get_velocity(self):
        if self._live():
            return np.array([float(self._cell('light_vx').item()), float(self._cell('light_vy').item())])
        return self._velocity

    def get_state(self):
        return np.r_[self.get_position(), self.get_velocity()]

    def step(self, action, time_step):
        vel = self.get_velocity().copy()
        if action is not None:
            a = np.asarray(action).squeeze()
            a = np.minimum(np.maximum(a, self._action_bounds[0]), self._action_bounds[1])
            vel = vel + a * time_step
        n = np.linalg.norm(vel)
        if n > self.max_velocity:
            vel = vel * (self.max_velocity / n)
        self._velocity = vel
        if self._live():
            self._put('light_vx', vel[0])
            self._put('light_vy', vel[1])
        pos = self.get_position() + vel * time_step
        self.set_position(np.minimum(np.maximum(pos, self._bounds[0]), self._bounds[1]))


class CompositeLight(Light):
    """light.py:99-148: value = sum of the component values, gradient = gradient of the brightest component.
    On the device the components must be CircularGradientLight / MomentumLight (at most 4)."""
    light_type = nat.LIGHT_COMPOSITE

    def __init__(self, lights=None, reducer=np.sum):
        super().__init__()
        self._lights = list(lights)
        self._reducer = reducer
        self.observation_space = Box(np.concatenate([l.observation_space.low for l in self._lights]),
                                     np.concatenate([l.observation_space.high for l in self._lights]), dtype=np.float64)
        self.action_space = Box(np.concatenate([l.action_space.low for l in self._lights]),
                                np.concatenate([l.action_space.high for l in self._lights]), dtype=np.float64)
        self._action_dims = [l.action_space.shape[0] for l in self._lights]

    @property
    def lights(self):
        return tuple(self._lights)

    def step(self, action, time_step):
        if action is not None:
            action = np.asarray(action).squeeze()
            for l, ad in zip(self._lights, self._action_dims):
                l.step(action[:ad], time_step)
                action = action[ad:]

    def get_value(self, position):
        return np.sum(np.array([l.get_value(position) for l in self._lights]), axis=0)

    def value_and_gradients(self, position):
        values, grads = map(np.asarray, zip(*[l.value_and_gradients(position) for l in self._lights]))
        value = np.sum(values, axis=0)
        max_l = np.argmax(values, axis=0)
        return value, grads[max_l, range(np.asarray(position).shape[0])].squeeze()

    def get_gradient(self, position):
        return self.value_and_gradients(position)[1]

    def get_state(self):
        return np.concatenate([l.get_state() for l in self._lights])


class SmoothGridLight(Light):
    """light.py:198-215: declared but not implemented in the reference (every method raises)."""

    def __init__(self):
        super().__init__()

    def step(self, action, time_step):
        raise NotImplementedError

    def get_value(self, position):
        raise NotImplementedError

    def get_gradient(self, position):
        raise NotImplementedError

    def get_state(self):
        raise NotImplementedError
