"""YAML-configured scenes on the batched device simulator.

Drop-in for the reference's `YamlKilobotsEnv` / `EnvConfiguration`
(gym_kilobots/envs/yaml_kilobots_env.py): the same YAML tags (`!EvalEnv`, `!ObjectConf`, `!LightConf`,
`!KilobotsConf`), the same constructor (`YamlKilobotsEnv(configuration=conf)`), the same spaces and
random-initialisation rules.  What runs on the GPU: every light model, kilobots of any drive law, every object shape
(circle, square / rect, corner_quad, triangle, l_shape, t_shape, c_shape) as long as one env holds at most 8 convex
fixtures in total (`UnknownObjectException` at reset() beyond that).
"""
import random

import numpy as np
import yaml

from .. import lib as kb_lib
from ..lib import (CForm, Circle, CircularGradientLight, CompositeLight, CornerQuad, GradientLight, LForm, MomentumLight,
                   Quad, SinglePositionLight, TForm, Triangle)
from ..spaces import Box
from .kilobots_env import KilobotsEnv, UnknownLightTypeException, UnknownObjectException


class _Record(yaml.YAMLObject):
    """Plain attribute bag with value equality (what the reference's configuration classes are)."""

    def __eq__(self, other):
        mine, theirs = vars(self), vars(other)
        return all(k in theirs and mine[k] == theirs[k] for k in mine)

    __hash__ = None


class EnvConfiguration(_Record):
    yaml_tag = '!EvalEnv'

    class ObjectConfiguration(_Record):
        yaml_tag = '!ObjectConf'

        def __init__(self, idx, color, shape, width, height, init, symmetry):
            self.idx, self.color, self.shape = idx, color, shape
            self.width, self.height, self.init, self.symmetry = width, height, init, symmetry

        @property
        def object_type(self):
            return 'square' if self.shape in ('corner_quad', 'corner-quad', 'quad') else self.shape

    class LightConfiguration(_Record):
        yaml_tag = '!LightConf'

        def __init__(self, obj_type, init, radius=None):
            self.type, self.init, self.radius = obj_type, init, radius

    class KilobotsConfiguration(_Record):
        yaml_tag = '!KilobotsConf'

        def __init__(self, num, mean, std, type='SimplePhototaxisKilobot'):
            self.num, self.mean, self.std, self.type = num, mean, std, type

    def __init__(self, width, height, resolution, objects, light, kilobots):
        self.width, self.height, self.resolution = width, height, resolution
        self.objects = [self.ObjectConfiguration(**o) for o in objects]
        self.light = self.LightConfiguration(**light)
        self.kilobots = self.KilobotsConfiguration(**kilobots)


def rot_matrix(alpha):
    c, s = np.cos(alpha), np.sin(alpha)
    return np.array([[c, -s], [s, c]])


# shape strings of the reference's object factory (yaml_kilobots_env.py:216-245)
_SHAPE_CLASSES = {'square': Quad, 'quad': Quad, 'rect': Quad, 'corner_quad': CornerQuad, 'corner-quad': CornerQuad,
                  'triangle': Triangle, 'l_shape': LForm, 't_shape': TForm, 'c_shape': CForm}


class YamlKilobotsEnv(KilobotsEnv):
    # the reference calls _init_kilobots() without the configured type, so `conf.kilobots.type` is ignored and
    # every kilobot is a SimplePhototaxisKilobot (yaml_kilobots_env.py:147,327); set True to honour the YAML.
    honour_kilobot_type = False

    def __new__(cls, *, configuration, **kwargs):
        # class-level world size, exactly like the reference (shared by all instances of the class)
        cls.world_width, cls.world_height = configuration.width, configuration.height
        cls.world_size = cls.world_width, cls.world_height
        cls.screen_width = int(configuration.resolution * configuration.width)
        cls.screen_height = int(configuration.resolution * configuration.height)
        cls.screen_size = cls.screen_width, cls.screen_width
        return super(YamlKilobotsEnv, cls).__new__(cls, **kwargs)

    def __init__(self, *, configuration, **kwargs):
        self.conf = configuration
        self._progress_factor = 1.
        self._iteration_counter = 0
        super().__init__(**kwargs)

    def __eq__(self, other):
        return self.conf == other.conf

    __hash__ = None

    # ------------------------------------------------------------------ bookkeeping properties
    @property
    def progress_factor(self):
        return self._progress_factor

    @progress_factor.setter
    def progress_factor(self, pf):
        assert .0 <= pf <= 1., 'progress_factor must be a value in the range [.0, 1.]'
        self._progress_factor = pf

    @property
    def iteration_counter(self):
        return self._iteration_counter

    @iteration_counter.setter
    def iteration_counter(self, ic):
        assert isinstance(ic, int) and 0 <= ic, 'iteration_counter must be a positive integer'
        self._iteration_counter = ic

    def inc_iteration_counter(self):
        self._iteration_counter += 1

    # ------------------------------------------------------------------ scene construction
    def _configure_environment(self):
        self._init_objects()
        self._init_light()
        self._init_kilobots()

    def _world_sample(self):
        return np.random.rand(2) * np.asarray(self.world_size) + self.world_bounds[0]

    def _get_random_object_init(self):
        xy = self._world_sample() * 0.7
        return np.r_[xy, np.random.rand() * 2 * np.pi - np.pi]

    def _init_objects(self):
        for o in self.conf.objects:
            self._init_object(o.shape, o.width, o.height, o.init, o.color)

    def _init_object(self, object_shape, object_width, object_height, object_init, object_color=None):
        if object_init == 'random':
            object_init = self._get_random_object_init()
        if object_shape == 'circle':
            # note: the reference passes the configured *width* as the radius (yaml_kilobots_env.py:229-231)
            obj = Circle(radius=object_width, position=object_init[:2], orientation=object_init[2], world=self.world)
        elif object_shape in _SHAPE_CLASSES:
            obj = _SHAPE_CLASSES[object_shape](width=object_width, height=object_height, position=object_init[:2],
                                               orientation=object_init[2], world=self.world)
        else:
            raise UnknownObjectException('Shape of form {} not known.'.format(object_shape))
        if object_color:
            obj.color = object_color
        self._add_object(obj)

    def _init_light(self):
        if hasattr(self.conf, 'light'):
            self._light = self._init_light_from_config(self.conf.light)

    def _get_random_light_init(self, at_object=False):
        if not at_object:
            return self._world_sample()
        target = self._objects[np.random.choice(len(self._objects), 1)[0]]
        reach = 1.2 * max(target.width, target.height) / 2
        phi = np.random.rand() * 2 * np.pi - np.pi
        return target.get_position() + (np.cos(phi) * reach, np.sin(phi) * reach)

    def _init_light_from_config(self, light_config):
        kind = light_config.type
        unit = np.array([1, 1]) * .01
        if kind in ('circular', 'momentum'):
            where = light_config.init
            if where == 'random':
                where = self._get_random_light_init()
            elif where == 'object':
                where = self._get_random_light_init(at_object=True)
            common = dict(position=np.asarray(where, dtype=np.float64), radius=light_config.radius,
                          bounds=tuple(np.array(self.world_bounds) * 1.1), action_bounds=(-unit, unit))
            if kind == 'circular':
                return CircularGradientLight(**common)
            heading = np.random.rand() * 2 * np.pi - np.pi
            return MomentumLight(velocity=np.array([np.sin(heading), np.cos(heading)]) * .01, max_velocity=.01, **common)
        if kind == 'linear':
            return GradientLight(angle=light_config.init)
        if kind == 'composite':
            if light_config.init == 'random':
                random.shuffle(light_config.components)
            return CompositeLight([self._init_light_from_config(c) for c in light_config.components])
        raise UnknownLightTypeException()

    def _init_kilobots(self, type='SimplePhototaxisKilobot'):
        spec = self.conf.kilobots
        if self.honour_kilobot_type:
            type = spec.type
        count, centre = spec.num, spec.mean
        if isinstance(centre, str) and centre == 'light':
            if isinstance(self._light, SinglePositionLight):
                centre = self._light.get_position()
            elif isinstance(self._light, CompositeLight):
                spots = np.asarray([l.get_position() for l in self._light.lights])
                centre = spots[np.random.choice(np.arange(len(spots)), count)]
            else:
                centre = 'random'
        if isinstance(centre, str) and centre == 'random':
            centre = self._world_sample() * 0.9
        # Gaussian cloud around the centre, clipped 2 cm inside the arena (yaml_kilobots_env.py:346-352)
        cloud = np.random.normal(scale=spec.std, size=(count, 2)) + centre
        lo, hi = self.world_bounds[0] + 0.02, self.world_bounds[1] - 0.02
        maker = getattr(kb_lib, type)
        for spot in np.clip(cloud, lo, hi):
            self._add_kilobot(maker(self.world, position=spot, light=self._light))

    # ------------------------------------------------------------------ spaces
    def _tiled_box(self, low, high, count):
        return Box(low=np.array(list(low) * count), high=np.array(list(high) * count), dtype=np.float64)

    @property
    def kilobots_state_space(self):
        return self._tiled_box((self.world_x_range[0], self.world_y_range[0]),
                               (self.world_x_range[1], self.world_y_range[1]), len(self._kilobots))

    kilobots_observation_space = kilobots_state_space

    @property
    def object_state_space(self):
        return self._tiled_box((self.world_x_range[0], self.world_y_range[0], -np.inf),
                               (self.world_x_range[1], self.world_y_range[1], np.inf), len(self._objects))

    @property
    def object_observation_space(self):
        # objects are observed as x, y, sin(theta), cos(theta)
        return self._tiled_box((self.world_x_range[0], self.world_y_range[0], -1., -1.),
                               (self.world_x_range[1], self.world_y_range[1], 1., 1.), len(self._objects))

    @property
    def light_state_space(self):
        return self._light.observation_space if self._light else None

    @property
    def light_observation_space(self):
        return self._light.observation_space if (self._light and self._observe_light) else None

    @property
    def action_space(self):
        return self._light.action_space if self._light else None

    def _joined_box(self, parts):
        parts = [p for p in parts if p]
        return Box(low=np.concatenate([p.low for p in parts]), high=np.concatenate([p.high for p in parts]),
                   dtype=np.float32)

    @property
    def state_space(self):
        return self._joined_box([self.kilobots_state_space, self.light_state_space, self.object_state_space])

    @property
    def observation_space(self):
        return self._joined_box([self.kilobots_state_space, self.light_observation_space, self.object_observation_space])

    def get_reward(self, state, action, new_state):
        return .0
