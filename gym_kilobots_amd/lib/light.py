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

    def _live(self):
        return self._world is not None and self._world.backend is not None

    def get_position(self):
        if self._live():
            be, e = self._world.backend, self._world.env_index
            return np.array([float(be.light_x[e].item()), float(be.light_y[e].item())])
        return self._position

    def set_position(self, position):
        self._position = np.asarray(position, dtype=np.float64)
        if self._live():
            be, e = self._world.backend, self._world.env_index
            be.light_x[e] = float(position[0])
            be.light_y[e] = float(position[1])

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
