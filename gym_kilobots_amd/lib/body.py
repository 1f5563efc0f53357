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
        self.version = 0        # bumped by every write to the device state that goes through the body / light views
        # host-programmed kilobots (KilobotsEnv._step_host_programmed): per-substep host copies of env 0's arrays, name -> [N]
        # numpy array; reads and writes of the body views go there while the kilobots' _loop runs
        self.host_cache = None
        self.host_dirty = set()

    def touch(self):
        self.version += 1

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
        hc = self._world.host_cache
        if hc is not None and name in hc:
            return float(hc[name][self._index])
        return float(getattr(self._world.backend, name)[self._world.env_index, self._index].item())

    def _set(self, name, value):
        hc = self._world.host_cache
        if hc is not None and name in hc:
            hc[name][self._index] = value
            self._world.host_dirty.add(name)
            return
        getattr(self._world.backend, name)[self._world.env_index, self._index] = float(value)
        self._world.touch()

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
        """True if a touching contact with `other` exists (reference body.py:87-90 walks Box2D's contact list).  Here the
        narrowphase predicates are evaluated on the current poses in fp32 world units: b2CollideCircles,
        b2CollidePolygonAndCircle, and for two polygons the separating-axis stage of b2CollidePolygons (skin radii
        included).  Returns True or None like the reference."""
        mine, theirs = self._world_fixtures(), other._world_fixtures()
        for fa in mine:
            for fb in theirs:
                if _fixtures_touch(fa, fb):
                    return True
        return None

    def _world_fixtures(self):
        """[(kind, centre or vertices [world units], radius)] of this body at its current pose."""
        x, y, th = self.get_pose()
        c, s = np.cos(th), np.sin(th)
        out = []
        for kind, radius, verts in self._shape_spec():
            if kind == 0:
                out.append(('circle', np.array([x, y]) * _world_scale, radius * _world_scale))
                continue
            if kind == 1:
                hx, hy = verts[0]
                verts = [(-hx, -hy), (hx, -hy), (hx, hy), (-hx, hy)]
            v = np.array(verts, dtype=np.float64)
            w = np.stack([c * v[:, 0] - s * v[:, 1], s * v[:, 0] + c * v[:, 1]], -1) + np.array([x, y]) * _world_scale
            out.append(('poly', w, 0.01))
        return out

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


def _poly_normals(v):
    e = np.roll(v, -1, axis=0) - v
    n = np.stack([e[:, 1], -e[:, 0]], -1)
    return n / np.linalg.norm(n, axis=1, keepdims=True)


def _fixtures_touch(fa, fb):
    (ka, pa, ra), (kb, pb, rb) = fa, fb
    if ka == 'circle' and kb == 'circle':
        d = np.float32(pb) - np.float32(pa)
        rr = np.float32(ra) + np.float32(rb)
        return bool(np.float32(d[0] * d[0]) + np.float32(d[1] * d[1]) <= np.float32(rr * rr))
    if ka == 'circle':
        return _fixtures_touch(fb, fa)
    na = _poly_normals(pa)
    if kb == 'circle':                                   # b2CollidePolygonAndCircle
        radius = ra + rb
        sep = ((pb - pa) * na).sum(1)
        i = int(sep.argmax())
        if sep[i] > radius:
            return False
        if sep[i] < 1.19209290e-07:
            return True
        v1, v2 = pa[i], pa[(i + 1) % len(pa)]
        u1, u2 = np.dot(pb - v1, v2 - v1), np.dot(pb - v2, v1 - v2)
        if u1 <= 0:
            return bool(np.dot(pb - v1, pb - v1) <= radius * radius)
        if u2 <= 0:
            return bool(np.dot(pb - v2, pb - v2) <= radius * radius)
        return True
    nb = _poly_normals(pb)                               # b2FindMaxSeparation both ways
    total = ra + rb
    sep_a = max(min(np.dot(n, q - v) for q in pb) for n, v in zip(na, pa))
    sep_b = max(min(np.dot(n, q - v) for q in pa) for n, v in zip(nb, pb))
    return bool(sep_a <= total and sep_b <= total)


class Circle(Body):
    def __init__(self, radius, **kwargs):
        super().__init__(**kwargs)
        self._radius = radius

    def _arrays(self):
        return ('ox', 'oy', 'otheta')

    def _shape_spec(self):
        """Fixtures as the device configuration takes them: a list of (kb_shape, radius [m], vertices [world units])."""
        return [(0, float(self._radius), [])]

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


class Quad(Body):
    """Box of width x height centred on the body origin (reference body.py:129-163: b2PolygonShape::SetAsBox)."""

    def __init__(self, width, height, **kwargs):
        super().__init__(**kwargs)
        self._width = width
        self._height = height

    def _arrays(self):
        return ('ox', 'oy', 'otheta')

    def _shape_spec(self):
        return [(1, 0.0, [[self._width / 2 * _world_scale, self._height / 2 * _world_scale]])]      # body.py:137

    @property
    def width(self):
        return self._width

    @property
    def height(self):
        return self._height

    def get_width(self):
        return self._width

    def get_height(self):
        return self._height

    @property
    def local_vertices(self):
        hx, hy = self._width / 2, self._height / 2
        return np.array([[(-hx, -hy), (hx, -hy), (hx, hy), (-hx, hy)]])          # SetAsBox order

    @property
    def vertices(self):
        return np.array([[self.get_world_point(v) for v in self.local_vertices[0]]])

    def plot(self, axes, **kwargs):
        from ..kb_plotting import plot_body
        return plot_body(axes, self, **kwargs)


class CornerQuad(Quad):
    """A Quad whose first three vertices are drawn highlighted (reference body.py:166-178); same physics."""


def _hull_order(points):
    """Vertex order b2PolygonShape::Set produces: gift wrapping, counter-clockwise, starting at the lowest of the
    right-most points (fp32 like Box2D)."""
    ps = [np.asarray(p, np.float32) for p in points]
    i0 = 0
    for i in range(1, len(ps)):
        if ps[i][0] > ps[i0][0] or (ps[i][0] == ps[i0][0] and ps[i][1] < ps[i0][1]):
            i0 = i
    hull, ih = [], i0
    while True:
        hull.append(ih)
        ie = 0
        for j in range(1, len(ps)):
            if ie == ih:
                ie = j
                continue
            r, v = ps[ie] - ps[hull[-1]], ps[j] - ps[hull[-1]]
            c = np.float32(r[0] * v[1]) - np.float32(r[1] * v[0])
            if c < 0 or (c == 0 and float(v @ v) > float(r @ r)):
                ie = j
        ih = ie
        if ie == i0:
            break
    return [points[i] for i in hull]


class Polygon(Body):
    """Polygon bodies of the reference (body.py:217-262): `_shape_vertices()` lists convex sub-polygons that are
    scaled to width x height and recentred on the area-weighted mean of their vertex means; every sub-polygon becomes
    one fixture of the body (at most 8 fixtures per env in total on the device)."""

    def __init__(self, width, height, **kwargs):
        super().__init__(**kwargs)
        self._width = width
        self._height = height
        vertices = np.array(self._shape_vertices(), dtype=np.float64)
        v_size = np.amax(vertices, (0, 1)) - np.amin(vertices, (0, 1))
        vertices /= v_size
        vertices *= np.array((width, height))
        centroid = np.zeros(2)
        area = .0
        for vs in vertices:
            a = 0.5 * np.abs(np.dot(vs[:, 0], np.roll(vs[:, 1], 1)) - np.dot(vs[:, 1], np.roll(vs[:, 0], 1)))
            area += a
            centroid += vs.mean(axis=0) * a
        centroid /= area
        self._local_vertices = vertices - centroid
        self._local_vertices.setflags(write=False)

    def _arrays(self):
        return ('ox', 'oy', 'otheta')

    def _shape_spec(self):
        if not all(3 <= len(vs) <= 4 for vs in self._local_vertices):
            raise NotImplementedError('%s: convex fixtures of 3 or 4 vertices run on the device' % type(self).__name__)
        return [(2, 0.0, [[float(x), float(y)] for x, y in _hull_order([tuple(v) for v in vs * _world_scale])])
                for vs in self._local_vertices]                                         # body.py:243-251

    @property
    def width(self):
        return self._width

    @property
    def height(self):
        return self._height

    @property
    def local_vertices(self):
        return self._local_vertices

    @property
    def vertices(self):
        return np.array([[self.get_world_point(v) for v in vs] for vs in self._local_vertices])

    @property
    def plot_vertices(self):
        raise NotImplementedError

    @staticmethod
    def _shape_vertices():
        raise NotImplementedError

    def plot(self, axes, **kwargs):
        from ..kb_plotting import plot_body
        return plot_body(axes, self, **kwargs)


class Triangle(Polygon):
    @staticmethod
    def _shape_vertices():
        return np.array([[(-0.5, 0.0), (0.0, 0.0), (0.0, 1.0)]])

    @property
    def plot_vertices(self):
        return self.vertices.reshape((-1, 2))


class LForm(Polygon):
    @staticmethod
    def _shape_vertices():
        return np.array([[(-0.05, 0.0), (0.1, 0.0), (0.1, 0.3), (-0.05, 0.3)],
                         [(0.1, 0.0), (0.1, -0.15), (-0.2, -0.15), (-0.2, 0.0)]])

    @property
    def plot_vertices(self):
        return self.vertices.reshape((-1, 2))[[0, 7, 6, 5, 2, 3], :]


class TForm(Polygon):
    @staticmethod
    def _shape_vertices():
        return np.array([[(0.0, 0.15), (0.2, 0.15), (0.2, -0.15), (0.0, -0.15)],
                         [(0.0, 0.05), (0.0, -0.05), (-0.2, -0.05), (-0.2, 0.05)]])

    @property
    def plot_vertices(self):
        return self.vertices.reshape((-1, 2))[[0, 1, 2, 3, 5, 6, 7, 4], :]


class CForm(Polygon):
    @staticmethod
    def _shape_vertices():
        return np.array([[(0.09, 0.15), (0.09, -0.15), (-0.01, -0.15), (-0.01, 0.15,)],
                         [(-0.01, -0.15), (-0.11, -0.15), (-0.11, -0.08), (-0.01, -0.05)],
                         [(-0.01, 0.15), (-0.11, 0.15), (-0.11, 0.08), (-0.01, 0.05)]])

    @property
    def plot_vertices(self):
        return self.vertices.reshape((-1, 2))[[0, 1, 5, 6, 7, 11, 10, 9], :]
