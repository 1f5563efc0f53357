"""Body views: the reference's world/body API (gym_kilobots/lib/body.py) on top of the batched
device state.  A view addresses body `index` of env `env_index` of a simulator backend; nothing is
simulated here.  Poses are metres at this level, the backend stores Box2D world units (x25)."""
import numpy as np

_world_scale = 25.


class World:
    """Stand-in for the `b2World` handle that reference body constructors take as first argument
    (body.py:18-38).  It only records the bodies created during `_configure_environment`; the env
    turns that list into device buffers (envs/kilobots_env.py)."""

    def __init__(self):
        self.kilobots = []
        self.objects = []
        self.backend = None
        self.env_index = 0

    def _register(self, body):
        from .kilobot import Kilobot
        if isinstance(body, Kilobot):
            body._index = len(self.kilobots)
            self.kilobots.append(body)
        else:
            body._index = len(self.objects)
            self.objects.append(body)

    def clear(self):
        del self.kilobots[:]
        del self.objects[:]


class Body:
    _density = 2
    _friction = 0.01
    _restitution = 0.0

    _linear_damping = .8
    _angular_damping = .8

    def __init__(self, world, position=None, orientation=None):
        if self.__class__ == Body:
            raise NotImplementedError('Abstract class Body cannot be instantiated.')
        self._color = np.array((93, 133, 195))
        self._highlight_color = np.array((238, 80, 62))
        if position is None:
            position = [.0, .0]
        if orientation is None:
            orientation = .0
        self._world = world
        self._index = -1
        # pose until the env has uploaded it (fp32 like b2Body)
        self._init_pose = (np.float32(_world_scale * float(position[0])) / _world_scale,
                           np.float32(_world_scale * float(position[1])) / _world_scale,
                           float(np.float32(orientation)))
        world._register(self)

    # ---- backend access -------------------------------------------------------------------
    def _arrays(self):
        raise NotImplementedError

    def _live(self):
        return self._world.backend is not None and self._index >= 0

    def _get(self, name):
        return float(getattr(self._world.backend, name)[self._world.env_index, self._index].item())

    def _set(self, name, value):
        getattr(self._world.backend, name)[self._world.env_index, self._index] = float(value)

    @property
    def width(self):
        raise NotImplementedError

    @property
    def height(self):
        raise NotImplementedError

    def get_position(self):
        x, y, _ = self.get_pose()
        return np.array([x, y])

    def set_position(self, position):
        if self._live():
            ax, ay, _ = self._arrays()
            self._set(ax, np.float32(float(position[0]) * _world_scale))
            self._set(ay, np.float32(float(position[1]) * _world_scale))
            self._world.backend.forget_contacts()
        else:
            self._init_pose = (float(position[0]), float(position[1]), self._init_pose[2])

    def get_orientation(self):
        return self.get_pose()[2]

    def set_orientation(self, orientation):
        if self._live():
            self._set(self._arrays()[2], orientation)
        else:
            self._init_pose = (self._init_pose[0], self._init_pose[1], float(orientation))

    def get_pose(self):
        if not self._live():
            return tuple(self._init_pose)
        ax, ay, at = self._arrays()
        return (self._get(ax) / _world_scale, self._get(ay) / _world_scale, self._get(at))

    def set_pose(self, pose):
        self.set_position(pose[:2])
        self.set_orientation(pose[2])

    def get_state(self):
        return self.get_pose()

    def get_local_point(self, point):
        # b2Body::GetLocalPoint: R(theta)^T (p - position)
        x, y, th = self.get_pose()
        dx, dy = float(point[0]) - x, float(point[1]) - y
        c, s = np.cos(th), np.sin(th)
        return np.array([c * dx + s * dy, -s * dx + c * dy])

    def get_local_orientation(self, angle):
        return angle - self.get_orientation()

    def get_local_pose(self, pose):
        return tuple((*self.get_local_point(pose[:2]), self.get_local_orientation(pose[2])))

    def get_world_point(self, point):
        x, y, th = self.get_pose()
        c, s = np.cos(th), np.sin(th)
        return np.array([c * point[0] - s * point[1] + x, s * point[0] + c * point[1] + y])

    def collides_with(self, other):
        """True if a touching contact with `other` exists (reference body.py:87-90 walks Box2D's
        contact list; here: the manifold test b2CollideCircles applies, distance <= rA + rB)."""
        ra, rb = getattr(self, '_radius', None), getattr(other, '_radius', None)
        if ra is None or rb is None:
            raise NotImplementedError('collides_with is implemented for circular bodies')
        (xa, ya, _), (xb, yb, _) = self.get_pose(), other.get_pose()
        # compare in fp32 world units like the kernel does
        dx = np.float32(xb * _world_scale) - np.float32(xa * _world_scale)
        dy = np.float32(yb * _world_scale) - np.float32(ya * _world_scale)
        rr = np.float32(ra * _world_scale) + np.float32(rb * _world_scale)
        if np.float32(dx * dx) + np.float32(dy * dy) <= np.float32(rr * rr):
            return True
        return None

    @property
    def color(self):
        return self._color

    @color.setter
    def color(self, color):
        self._color = np.clip(np.asarray(color, dtype=np.int32), 0, 255)

    @property
    def highlight_color(self):
        return self._highlight_color

    @highlight_color.setter
    def highlight_color(self, color):
        self._highlight_color = np.clip(np.asarray(color, dtype=np.int32), 0, 255)

    def draw(self, viewer):
        raise NotImplementedError('rendering is outside the accelerated hot path (SURVEY.md 2, component 8)')

    def plot(self, axes, **kwargs):
        raise NotImplementedError('plotting is outside the accelerated hot path (SURVEY.md 2, component 9)')


class Circle(Body):
    def __init__(self, radius, **kwargs):
        super().__init__(**kwargs)
        self._radius = radius

    def _arrays(self):
        return ('ox', 'oy', 'otheta')

    @property
    def width(self):
        return 2 * self._radius

    @property
    def height(self):
        return 2 * self._radius

    @property
    def vertices(self):
        return np.array([[self.get_position()]])

    def get_radius(self):
        return self._radius
