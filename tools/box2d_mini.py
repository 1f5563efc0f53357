#!/usr/bin/env python3
"""box2d_mini -- an INDEPENDENT float32 restatement of Box2D 2.3.1's discrete step for a handful of bodies.

Purpose: oracle/kb_oracle.c and the HIP kernel are twins of ONE restatement of Box2D, so a
transcription error shared by both is invisible to the bit-exact GPU tests.  This file is a second, separate derivation,
written from Box2D's published algorithm in Box2D's own structure (b2Body / b2Fixture / b2Contact / b2Island /
b2ContactSolver, objects and lists, contacts solved in creation order) -- NOT from kb_oracle.c, which is organised around
grids, keys and packed arrays.  tools/gen_mini_solver_golden.py runs small scenes through it and commits the trajectories
as tests/golden/mini_solver.json; tests/test_oracle_vs_mini_solver.py holds the oracle to them.

It is NOT Box2D: box2d-py cannot be installed here, so this is still a restatement and row a10 stays "parity unpinned".
What it buys is independence: two derivations agreeing on contact manifolds, effective masses, the friction / block
solver, warm starting by feature id, position correction and the integrator.

Scope: b2World::Step(dt, velIters, posIters) with allowSleep = False, no gravity, no joints; continuousPhysics optional
(World(continuous=True)): b2World::SolveTOI for CIRCLES against the static edges (non-bullet bodies only meet static bodies
there): b2TimeOfImpact with the closed-form circle-centre / edge separation, b2Body::Advance, the TOI mini-island with
b2ContactSolver::SolveTOIPositionConstraints (Baumgarte 0.75, 20 iterations), a velocity solve without warm starting and the
integration of the rest of the step; polygons take no part in the continuous step here.  Discrete step:
  b2CollideCircles, b2CollidePolygonAndCircle, b2CollidePolygons, b2CollideEdgeAndCircle, b2CollideEdgeAndPolygon for a
  lone edge (no ghost vertices), b2Contact::Update (impulses carried over by feature id), b2Island::Solve (damping model
  selectable: Pade of 2.3.1 or the clamped linear form of 2.3.0), b2ContactSolver (friction, restitution threshold,
  2-point block solver with its four cases), position solver with Baumgarte 0.2, translation / rotation clamps.
All arithmetic is numpy.float32, one rounding per operation, in the operation order of the C++ source as far as the
author recalls it; comparisons with the oracle therefore use a tolerance, not bit equality.
"""
import math

import numpy as np

f32 = np.float32
PI = f32(3.14159265359)
LINEAR_SLOP = f32(0.005)
ANGULAR_SLOP = f32(2.0 / 180.0) * PI
POLYGON_RADIUS = f32(2.0) * LINEAR_SLOP
MAX_MANIFOLD_POINTS = 2
VELOCITY_THRESHOLD = f32(1.0)
BAUMGARTE = f32(0.2)
MAX_LINEAR_CORRECTION = f32(0.2)
MAX_TRANSLATION = f32(2.0)
MAX_ROTATION = f32(0.5) * PI
TIME_TO_SLEEP = f32(0.5)                                     # b2_timeToSleep
LINEAR_SLEEP_TOLERANCE = f32(0.01)                           # b2_linearSleepTolerance
ANGULAR_SLEEP_TOLERANCE = f32(2.0) / f32(180.0) * PI         # b2_angularSleepTolerance
EPSILON = f32(1.192092896e-07)
TOI_BAUMGARTE = f32(0.75)
MAX_SUB_STEPS = 8
MAX_TOI_CONTACTS = 32
MAXFLOAT = f32(3.402823466e+38)


# ---------------------------------------------------------------------------------------------- b2Math
class V:
    """b2Vec2 with float32 components."""
    __slots__ = ('x', 'y')

    def __init__(self, x=0.0, y=0.0):
        self.x, self.y = f32(x), f32(y)

    def __add__(self, o): return V(self.x + o.x, self.y + o.y)
    def __sub__(self, o): return V(self.x - o.x, self.y - o.y)
    def __neg__(self): return V(-self.x, -self.y)
    def __rmul__(self, s): return V(f32(s) * self.x, f32(s) * self.y)
    def copy(self): return V(self.x, self.y)
    def length(self): return f32(np.sqrt(self.x * self.x + self.y * self.y))
    def length_sq(self): return self.x * self.x + self.y * self.y

    def normalized(self):
        """b2Vec2::Normalize (returns the vector; zero below epsilon stays as it is)."""
        ln = self.length()
        if ln < EPSILON:
            return self.copy()
        inv = f32(1.0) / ln
        return V(self.x * inv, self.y * inv)

    def tup(self): return (float(self.x), float(self.y))


def dot(a, b): return a.x * b.x + a.y * b.y
def cross(a, b): return a.x * b.y - a.y * b.x
def cross_vs(a, s): return V(f32(s) * a.y, -f32(s) * a.x)
def cross_sv(s, a): return V(-f32(s) * a.y, f32(s) * a.x)
def dist_sq(a, b): return (a - b).length_sq()


def clamp(a, lo, hi): return max(f32(lo), min(f32(a), f32(hi)))


class Rot:
    __slots__ = ('s', 'c')

    def __init__(self, angle=0.0):
        # Box2D calls sinf / cosf; float32 results of the double-precision functions differ from a correctly rounded
        # single-precision libm by at most 1 ulp
        self.s, self.c = f32(math.sin(float(f32(angle)))), f32(math.cos(float(f32(angle))))


class XF:
    """b2Transform"""
    __slots__ = ('p', 'q')

    def __init__(self, p=None, q=None):
        self.p, self.q = p if p is not None else V(), q if q is not None else Rot()


def rot_mul(q, v): return V(q.c * v.x - q.s * v.y, q.s * v.x + q.c * v.y)
def rot_mulT(q, v): return V(q.c * v.x + q.s * v.y, -q.s * v.x + q.c * v.y)
def xf_mul(t, v): return V((t.q.c * v.x - t.q.s * v.y) + t.p.x, (t.q.s * v.x + t.q.c * v.y) + t.p.y)


def xf_mulT(t, v):
    px, py = v.x - t.p.x, v.y - t.p.y
    return V(t.q.c * px + t.q.s * py, -t.q.s * px + t.q.c * py)


def xf_mulT_xf(A, B):
    """b2MulT(A, B) = inverse(A) * B"""
    C = XF()
    C.q = Rot()
    C.q.s = A.q.c * B.q.s - A.q.s * B.q.c
    C.q.c = A.q.c * B.q.c + A.q.s * B.q.s
    C.p = rot_mulT(A.q, B.p - A.p)
    return C


# ---------------------------------------------------------------------------------------------- shapes
class Circle:
    kind = 'circle'

    def __init__(self, radius, p=None):
        self.radius, self.p = f32(radius), p if p is not None else V()

    def compute_mass(self, density):
        mass = f32(density) * PI * self.radius * self.radius
        inertia = mass * (f32(0.5) * self.radius * self.radius + dot(self.p, self.p))
        return mass, self.p.copy(), inertia


class Polygon:
    kind = 'polygon'

    def __init__(self):
        self.radius = POLYGON_RADIUS
        self.vertices, self.normals, self.centroid = [], [], V()

    @staticmethod
    def box(hx, hy):
        """b2PolygonShape::SetAsBox"""
        p = Polygon()
        hx, hy = f32(hx), f32(hy)
        p.vertices = [V(-hx, -hy), V(hx, -hy), V(hx, hy), V(-hx, hy)]
        p.normals = [V(0.0, -1.0), V(1.0, 0.0), V(0.0, 1.0), V(-1.0, 0.0)]
        return p

    @staticmethod
    def hull(points):
        """b2PolygonShape::Set for points that already form a counter-clockwise convex hull."""
        p = Polygon()
        p.vertices = [V(x, y) for x, y in points]
        n = len(p.vertices)
        for i in range(n):
            edge = p.vertices[(i + 1) % n] - p.vertices[i]
            p.normals.append(cross_vs(edge, 1.0).normalized())
        p.centroid = _centroid(p.vertices)
        return p

    def compute_mass(self, density):
        """b2PolygonShape::ComputeMass: triangle fan about the average vertex."""
        n = len(self.vertices)
        center, area, I = V(), f32(0.0), f32(0.0)
        s = V()
        for v in self.vertices:
            s = s + v
        s = (f32(1.0) / f32(n)) * s
        k_inv3 = f32(1.0 / 3.0)
        for i in range(n):
            e1 = self.vertices[i] - s
            e2 = self.vertices[(i + 1) % n] - s
            D = cross(e1, e2)
            tri = f32(0.5) * D
            area = area + tri
            center = center + (tri * k_inv3) * (e1 + e2)
            intx2 = e1.x * e1.x + e2.x * e1.x + e2.x * e2.x
            inty2 = e1.y * e1.y + e2.y * e1.y + e2.y * e2.y
            I = I + (f32(0.25) * k_inv3 * D) * (intx2 + inty2)
        mass = f32(density) * area
        center = (f32(1.0) / area) * center
        c = center + s
        inertia = f32(density) * I
        inertia = inertia + mass * (dot(c, c) - dot(center, center))
        return mass, c, inertia


def _centroid(vs):
    n = len(vs)
    c, area = V(), f32(0.0)
    pref = V()
    inv3 = f32(1.0 / 3.0)
    for i in range(n):
        p1, p2, p3 = pref, vs[i], vs[(i + 1) % n]
        D = cross(p2 - p1, p3 - p1)
        tri = f32(0.5) * D
        area = area + tri
        c = c + (tri * inv3) * (p1 + p2 + p3)
    return (f32(1.0) / area) * c


class Edge:
    """b2EdgeShape without ghost vertices (a lone edge; static)."""
    kind = 'edge'

    def __init__(self, v1, v2):
        self.radius = POLYGON_RADIUS
        self.v1, self.v2 = V(*v1), V(*v2)

    def compute_mass(self, density):
        return f32(0.0), f32(0.5) * (self.v1 + self.v2), f32(0.0)


# ---------------------------------------------------------------------------------------------- bodies, fixtures
class Fixture:
    def __init__(self, body, shape, density, friction, restitution):
        self.body, self.shape = body, shape
        self.density, self.friction, self.restitution = f32(density), f32(friction), f32(restitution)


class Body:
    def __init__(self, world, position=(0.0, 0.0), angle=0.0, dynamic=True, linear_damping=0.0, angular_damping=0.0):
        self.world, self.dynamic = world, dynamic
        self.xf = XF(V(*position), Rot(angle))
        self.local_center = V()
        self.c0 = self.c = self.xf.p.copy()
        self.a0 = self.a = f32(angle)
        self.v, self.w = V(), f32(0.0)
        self.linear_damping, self.angular_damping = f32(linear_damping), f32(angular_damping)
        self.mass = self.inv_mass = f32(0.0)
        self.I = self.inv_I = f32(0.0)
        self.fixtures = []
        self.island_index = -1
        self.alpha0 = f32(0.0)
        self.awake, self.sleep_time = True, f32(0.0)       # b2BodyDef::awake = true, m_sleepTime = 0

    def set_awake(self, flag):
        """b2Body::SetAwake (2.3.x): waking an awake body does not touch its sleep time"""
        if flag:
            if not self.awake:
                self.awake, self.sleep_time = True, f32(0.0)
        else:
            self.awake, self.sleep_time = False, f32(0.0)
            self.v, self.w = V(), f32(0.0)

    def set_linear_velocity(self, v):
        """b2Body::SetLinearVelocity (what pybox2d's `body.linearVelocity = v` calls)"""
        if not self.dynamic:
            return
        if dot(v, v) > f32(0.0):
            self.set_awake(True)
        self.v = v

    def set_angular_velocity(self, w):
        """b2Body::SetAngularVelocity"""
        if not self.dynamic:
            return
        w = f32(w)
        if w * w > f32(0.0):
            self.set_awake(True)
        self.w = w

    def advance(self, alpha):
        """b2Body::Advance -> b2Sweep::Advance + SynchronizeTransform"""
        beta = (f32(alpha) - self.alpha0) / (f32(1.0) - self.alpha0)
        self.c0 = self.c0 + beta * (self.c - self.c0)
        self.a0 = self.a0 + beta * (self.a - self.a0)
        self.alpha0 = f32(alpha)
        self.c, self.a = self.c0.copy(), self.a0
        self.synchronize_transform()

    def create_fixture(self, shape, density=0.0, friction=0.2, restitution=0.0):
        f = Fixture(self, shape, density, friction, restitution)
        self.fixtures.append(f)
        self.reset_mass_data()
        self.world.new_fixture(f)
        return f

    def reset_mass_data(self):
        """b2Body::ResetMassData"""
        self.mass = self.inv_mass = self.I = self.inv_I = f32(0.0)
        self.local_center = V()
        if not self.dynamic:
            self.c0 = self.c = self.xf.p.copy()
            self.a0 = self.a
            return
        lc = V()
        for f in self.fixtures:
            if f.density == 0.0:
                continue
            m, c, i = f.shape.compute_mass(f.density)
            self.mass = self.mass + m
            lc = lc + m * c
            self.I = self.I + i
        if self.mass > 0.0:
            self.inv_mass = f32(1.0) / self.mass
            lc = self.inv_mass * lc
        else:
            self.mass = self.inv_mass = f32(1.0)
        if self.I > 0.0:
            self.I = self.I - self.mass * dot(lc, lc)
            self.inv_I = f32(1.0) / self.I
        else:
            self.I = self.inv_I = f32(0.0)
        old = self.c
        self.local_center = lc
        self.c0 = self.c = xf_mul(self.xf, lc)
        self.v = self.v + cross_sv(self.w, self.c - old)

    def synchronize_transform(self):
        self.xf.q = Rot(self.a)
        self.xf.p = self.c - rot_mul(self.xf.q, self.local_center)

    @property
    def position(self): return self.xf.p
    @property
    def angle(self): return self.a


# ---------------------------------------------------------------------------------------------- collision
class MPoint:
    __slots__ = ('local_point', 'normal_impulse', 'tangent_impulse', 'id')

    def __init__(self, lp, fid):
        self.local_point, self.id = lp, fid
        self.normal_impulse = self.tangent_impulse = f32(0.0)


class Manifold:
    def __init__(self):
        self.type, self.points = None, []          # 'circles' | 'faceA' | 'faceB'
        self.local_normal, self.local_point = V(), V()


def collide_circles(A, xfA, B, xfB):
    m = Manifold()
    pA, pB = xf_mul(xfA, A.p), xf_mul(xfB, B.p)
    d = pB - pA
    r = A.radius + B.radius
    if dot(d, d) > r * r:
        return m
    m.type, m.local_point, m.local_normal = 'circles', A.p.copy(), V()
    m.points = [MPoint(B.p.copy(), 0)]
    return m


def collide_polygon_circle(P, xfA, C, xfB):
    m = Manifold()
    c = xf_mul(xfB, C.p)
    cl = xf_mulT(xfA, c)
    normal_index, separation = 0, -MAXFLOAT
    radius = P.radius + C.radius
    n = len(P.vertices)
    for i in range(n):
        s = dot(P.normals[i], cl - P.vertices[i])
        if s > radius:
            return m
        if s > separation:
            separation, normal_index = s, i
    v1, v2 = P.vertices[normal_index], P.vertices[(normal_index + 1) % n]
    if separation < EPSILON:
        m.type, m.local_normal = 'faceA', P.normals[normal_index].copy()
        m.local_point = f32(0.5) * (v1 + v2)
        m.points = [MPoint(C.p.copy(), 0)]
        return m
    u1, u2 = dot(cl - v1, v2 - v1), dot(cl - v2, v1 - v2)
    if u1 <= 0.0:
        if dist_sq(cl, v1) > radius * radius:
            return m
        m.type, m.local_normal, m.local_point = 'faceA', (cl - v1).normalized(), v1.copy()
    elif u2 <= 0.0:
        if dist_sq(cl, v2) > radius * radius:
            return m
        m.type, m.local_normal, m.local_point = 'faceA', (cl - v2).normalized(), v2.copy()
    else:
        face_center = f32(0.5) * (v1 + v2)
        if dot(cl - face_center, P.normals[normal_index]) > radius:
            return m
        m.type, m.local_normal, m.local_point = 'faceA', P.normals[normal_index].copy(), face_center
    m.points = [MPoint(C.p.copy(), 0)]
    return m


def _find_max_separation(p1, xf1, p2, xf2):
    """b2FindMaxSeparation of Box2D 2.3.1: every edge normal of p1 against every vertex of p2."""
    xf = xf_mulT_xf(xf2, xf1)
    best, max_sep = 0, -MAXFLOAT
    for i in range(len(p1.vertices)):
        n = rot_mul(xf.q, p1.normals[i])
        v1 = xf_mul(xf, p1.vertices[i])
        si = MAXFLOAT
        for v2 in p2.vertices:
            sij = dot(n, v2 - v1)
            if sij < si:
                si = sij
        if si > max_sep:
            max_sep, best = si, i
    return best, max_sep


def _feature(ia, ib, ta, tb):
    """b2ContactFeature as a key: indexA, indexB, typeA, typeB (0 vertex, 1 face)"""
    return (ia, ib, ta, tb)


def _clip(v_in, normal, offset, vertex_index_a):
    """b2ClipSegmentToLine; v_in: [(point, id), (point, id)]"""
    out = []
    d0 = dot(normal, v_in[0][0]) - offset
    d1 = dot(normal, v_in[1][0]) - offset
    if d0 <= 0.0:
        out.append(v_in[0])
    if d1 <= 0.0:
        out.append(v_in[1])
    if d0 * d1 < 0.0:
        interp = d0 / (d0 - d1)
        p = v_in[0][0] + interp * (v_in[1][0] - v_in[0][0])
        out.append((p, _feature(vertex_index_a, v_in[0][1][1], 0, 1)))
    return out


def collide_polygons(A, xfA, B, xfB):
    m = Manifold()
    total = A.radius + B.radius
    eA, sA = _find_max_separation(A, xfA, B, xfB)
    if sA > total:
        return m
    eB, sB = _find_max_separation(B, xfB, A, xfA)
    if sB > total:
        return m
    k_tol = f32(0.1) * LINEAR_SLOP
    if sB > sA + k_tol:
        p1, p2, xf1, xf2, edge1, flip, m.type = B, A, xfB, xfA, eB, True, 'faceB'
    else:
        p1, p2, xf1, xf2, edge1, flip, m.type = A, B, xfA, xfB, eA, False, 'faceA'
    # b2FindIncidentEdge
    n1 = rot_mulT(xf2.q, rot_mul(xf1.q, p1.normals[edge1]))
    index, min_dot = 0, MAXFLOAT
    for i in range(len(p2.vertices)):
        d = dot(n1, p2.normals[i])
        if d < min_dot:
            min_dot, index = d, i
    i1, i2 = index, (index + 1) % len(p2.vertices)
    incident = [(xf_mul(xf2, p2.vertices[i1]), _feature(edge1, i1, 1, 0)),
                (xf_mul(xf2, p2.vertices[i2]), _feature(edge1, i2, 1, 0))]
    iv1, iv2 = edge1, (edge1 + 1) % len(p1.vertices)
    v11, v12 = p1.vertices[iv1], p1.vertices[iv2]
    local_tangent = (v12 - v11).normalized()
    local_normal = cross_vs(local_tangent, 1.0)
    plane_point = f32(0.5) * (v11 + v12)
    tangent = rot_mul(xf1.q, local_tangent)
    normal = cross_vs(tangent, 1.0)
    v11w, v12w = xf_mul(xf1, v11), xf_mul(xf1, v12)
    front_offset = dot(normal, v11w)
    side1 = -dot(tangent, v11w) + total
    side2 = dot(tangent, v12w) + total
    c1 = _clip(incident, -tangent, side1, iv1)
    if len(c1) < 2:
        m.type = None
        return m
    c2 = _clip(c1, tangent, side2, iv2)
    if len(c2) < 2:
        m.type = None
        return m
    m.local_normal, m.local_point = local_normal, plane_point
    for p, fid in c2[:MAX_MANIFOLD_POINTS]:
        if dot(normal, p) - front_offset <= total:
            if flip:
                fid = (fid[1], fid[0], fid[3], fid[2])
            m.points.append(MPoint(xf_mulT(xf2, p), fid))
    if not m.points:
        m.type = None
    return m


def collide_edge_circle(E, xfA, C, xfB):
    """b2CollideEdgeAndCircle for an edge without ghost vertices."""
    m = Manifold()
    Q = xf_mulT(xfA, xf_mul(xfB, C.p))
    A, B = E.v1, E.v2
    e = B - A
    u, v = dot(e, B - Q), dot(e, Q - A)
    radius = E.radius + C.radius
    if v <= 0.0:
        P = A
        d = Q - P
        if dot(d, d) > radius * radius:
            return m
        m.type, m.local_normal, m.local_point = 'circles', V(), P.copy()
        m.points = [MPoint(C.p.copy(), _feature(0, 0, 0, 0))]
        return m
    if u <= 0.0:
        P = B
        d = Q - P
        if dot(d, d) > radius * radius:
            return m
        m.type, m.local_normal, m.local_point = 'circles', V(), P.copy()
        m.points = [MPoint(C.p.copy(), _feature(1, 0, 0, 0))]
        return m
    den = dot(e, e)
    P = (f32(1.0) / den) * (u * A + v * B)
    d = Q - P
    if dot(d, d) > radius * radius:
        return m
    n = V(-e.y, e.x)
    if dot(n, Q - A) < 0.0:
        n = V(-n.x, -n.y)
    n = n.normalized()
    m.type, m.local_normal, m.local_point = 'faceA', n, A.copy()
    m.points = [MPoint(C.p.copy(), _feature(0, 0, 1, 0))]
    return m


def collide_edge_polygon(E, xfA, PB, xfB):
    """b2CollideEdgeAndPolygon (b2EPCollider::Collide) for an edge without ghost vertices."""
    m = Manifold()
    xf = xf_mulT_xf(xfA, xfB)
    centroidB = xf_mul(xf, PB.centroid)
    v1, v2 = E.v1, E.v2
    edge1 = (v2 - v1).normalized()
    normal1 = V(edge1.y, -edge1.x)
    offset1 = dot(normal1, centroidB - v1)
    front = offset1 >= 0.0
    if front:
        normal, lower, upper = normal1, -normal1, -normal1
    else:
        normal, lower, upper = -normal1, normal1, normal1
    pv = [xf_mul(xf, v) for v in PB.vertices]
    pn = [rot_mul(xf.q, n) for n in PB.normals]
    radius = f32(2.0) * POLYGON_RADIUS
    # edge axis
    e_sep = MAXFLOAT
    for v in pv:
        s = dot(normal, v - v1)
        if s < e_sep:
            e_sep = s
    if e_sep > radius:
        return m
    # polygon axis
    p_type, p_index, p_sep = None, -1, -MAXFLOAT
    perp = V(-normal.y, normal.x)
    for i in range(len(pv)):
        n = -pn[i]
        s1, s2 = dot(n, pv[i] - v1), dot(n, pv[i] - v2)
        s = min(s1, s2)
        if s > radius:
            p_type, p_index, p_sep = 'edgeB', i, s
            break
        if dot(n, perp) >= 0.0:
            if dot(n - upper, normal) < -ANGULAR_SLOP:
                continue
        else:
            if dot(n - lower, normal) < -ANGULAR_SLOP:
                continue
        if s > p_sep:
            p_type, p_index, p_sep = 'edgeB', i, s
    if p_type is not None and p_sep > radius:
        return m
    k_rel, k_abs = f32(0.98), f32(0.001)
    if p_type is None:
        primary = ('edgeA', 0 if front else 1)
    elif p_sep > k_rel * e_sep + k_abs:
        primary = ('edgeB', p_index)
    else:
        primary = ('edgeA', 0 if front else 1)
    nB = len(pv)
    if primary[0] == 'edgeA':
        m.type = 'faceA'
        best, best_v = 0, dot(normal, pn[0])
        for i in range(1, nB):
            val = dot(normal, pn[i])
            if val < best_v:
                best_v, best = val, i
        i1, i2 = best, (best + 1) % nB
        ie = [(pv[i1], _feature(0, i1, 1, 0)), (pv[i2], _feature(0, i2, 1, 0))]
        if front:
            rf_i1, rf_i2, rf_v1, rf_v2, rf_normal = 0, 1, v1, v2, normal1
        else:
            rf_i1, rf_i2, rf_v1, rf_v2, rf_normal = 1, 0, v2, v1, -normal1
    else:
        m.type = 'faceB'
        ie = [(v1, _feature(0, primary[1], 0, 1)), (v2, _feature(0, primary[1], 0, 1))]
        rf_i1, rf_i2 = primary[1], (primary[1] + 1) % nB
        rf_v1, rf_v2, rf_normal = pv[rf_i1], pv[rf_i2], pn[rf_i1]
    side_n1 = V(rf_normal.y, -rf_normal.x)
    side_n2 = -side_n1
    side_o1, side_o2 = dot(side_n1, rf_v1), dot(side_n2, rf_v2)
    c1 = _clip(ie, side_n1, side_o1, rf_i1)
    if len(c1) < MAX_MANIFOLD_POINTS:
        m.type = None
        return m
    c2 = _clip(c1, side_n2, side_o2, rf_i2)
    if len(c2) < MAX_MANIFOLD_POINTS:
        m.type = None
        return m
    if primary[0] == 'edgeA':
        m.local_normal, m.local_point = rf_normal.copy(), rf_v1.copy()
    else:
        m.local_normal, m.local_point = PB.normals[rf_i1].copy(), PB.vertices[rf_i1].copy()
    for p, fid in c2[:MAX_MANIFOLD_POINTS]:
        if dot(rf_normal, p - rf_v1) <= radius:
            if primary[0] == 'edgeA':
                m.points.append(MPoint(xf_mulT(xf, p), fid))
            else:
                m.points.append(MPoint(p.copy(), (fid[1], fid[0], fid[3], fid[2])))
    if not m.points:
        m.type = None
    return m


def _evaluate(fA, fB):
    """b2Contact type table; returns (fixtureA, fixtureB, manifold) with Box2D's ordering of the two shapes."""
    order = {'circle': 0, 'edge': 1, 'polygon': 2}
    a, b = fA, fB
    ka, kb = a.shape.kind, b.shape.kind
    if order[ka] > order[kb] or (ka == 'circle' and kb != 'circle'):
        # b2Contact::Create swaps so that s_registers[type1][type2] is a primary entry: (polygon, circle), (edge, circle),
        # (edge, polygon), (circle, circle), (polygon, polygon)
        pass
    table = {('circle', 'circle'): collide_circles, ('polygon', 'circle'): collide_polygon_circle,
             ('polygon', 'polygon'): collide_polygons, ('edge', 'circle'): collide_edge_circle,
             ('edge', 'polygon'): collide_edge_polygon}
    if (ka, kb) not in table:
        a, b = b, a
        ka, kb = kb, ka
    fn = table[(ka, kb)]
    return a, b, fn


class Contact:
    def __init__(self, fA, fB):
        self.fA, self.fB, self.fn = _evaluate(fA, fB)
        self.manifold = Manifold()
        self.touching = False
        self.toi, self.toi_valid, self.toi_count = f32(1.0), False, 0
        self.friction = f32(np.sqrt(self.fA.friction * self.fB.friction))          # b2MixFriction
        self.restitution = max(self.fA.restitution, self.fB.restitution)           # b2MixRestitution

    def update(self):
        """b2Contact::Update: new manifold, impulses of points with a matching id carried over."""
        old = self.manifold
        bA, bB = self.fA.body, self.fB.body
        new = self.fn(self.fA.shape, bA.xf, self.fB.shape, bB.xf)
        if new.type is None:
            new.points = []
        was_touching = self.touching
        self.touching = len(new.points) > 0
        if self.touching != was_touching:          # b2Contact::Update: "if (touching != wasTouching) { bodyA->SetAwake(true); bodyB->SetAwake(true); }"
            bA.set_awake(True)
            bB.set_awake(True)
        for mp in new.points:
            for op in old.points:
                if op.id == mp.id:
                    mp.normal_impulse, mp.tangent_impulse = op.normal_impulse, op.tangent_impulse
                    break
        self.manifold = new


# ---------------------------------------------------------------------------------------------- world
class World:
    """b2World with gravity (0, 0), warmStarting True; allow_sleep = b2World's doSleep, continuous = continuousPhysics.
    damping: 'pade' (Box2D >= 2.3.1) or 'linear' (Box2D <= 2.3.0)."""

    def __init__(self, damping='pade', continuous=False, allow_sleep=False):
        self.bodies, self.contacts, self.fixtures = [], [], []
        self.damping = damping
        self.continuous = continuous
        self.allow_sleep = allow_sleep
        self.inv_dt0 = f32(0.0)

    def create_body(self, **kw):
        b = Body(self, **kw)
        self.bodies.append(b)
        return b

    def new_fixture(self, f):
        # every pair of fixtures on different bodies with at least one dynamic body gets a contact, in creation order
        # (the broadphase only decides WHEN Box2D creates it; a contact without manifold points does nothing)
        for g in self.fixtures:
            if g.body is f.body or not (g.body.dynamic or f.body.dynamic):
                continue
            self.contacts.append(Contact(g, f))
        self.fixtures.append(f)

    def step(self, dt, vel_iters, pos_iters):
        dt = f32(dt)
        for c in self.contacts:          # b2ContactManager::Collide: contacts without an awake dynamic body are not updated
            bA, bB = c.fA.body, c.fB.body
            if not ((bA.dynamic and bA.awake) or (bB.dynamic and bB.awake)):
                continue
            c.update()
        dt_ratio = self.inv_dt0 * dt
        self._solve(dt, dt_ratio, vel_iters, pos_iters)
        if self.continuous:
            self._solve_toi(dt, vel_iters)
        self.inv_dt0 = f32(1.0) / dt

    # ------------------------------------------------------------------ b2World::SolveTOI (circles against static edges)
    def _solve_toi(self, dt, vel_iters):
        for b in self.bodies:
            b.alpha0 = f32(0.0)
        for c in self.contacts:
            c.toi, c.toi_valid, c.toi_count = f32(1.0), False, 0
        while True:
            min_c, min_alpha = None, f32(1.0)
            for c in self.contacts:
                if c.toi_count > MAX_SUB_STEPS:
                    continue
                if c.toi_valid:
                    alpha = c.toi
                else:
                    bA, bB = c.fA.body, c.fB.body
                    # non-bullet dynamic bodies only have TOI events with non-dynamic bodies
                    if (bA.dynamic and bB.dynamic) or not (bA.dynamic or bB.dynamic):
                        continue
                    if not ((bA.dynamic and bA.awake) or (bB.dynamic and bB.awake)):
                        continue                                   # b2World::SolveTOI: "is there a reason to collide?"
                    if not (c.fA.shape.kind == 'edge' and c.fB.shape.kind == 'circle'):
                        continue                                   # (polygons: not covered by this restatement)
                    alpha0 = max(bA.alpha0, bB.alpha0)
                    if bA.alpha0 < alpha0:
                        bA.advance(alpha0)
                    elif bB.alpha0 < alpha0:
                        bB.advance(alpha0)
                    state, beta = time_of_impact_edge_circle(c.fA.shape, bA, c.fB.shape, bB)
                    alpha = min(alpha0 + (f32(1.0) - alpha0) * beta, f32(1.0)) if state == 'touching' else f32(1.0)
                    c.toi, c.toi_valid = alpha, True
                if alpha < min_alpha:
                    min_c, min_alpha = c, alpha
            if min_c is None or f32(1.0) - f32(10.0) * EPSILON < min_alpha:
                break
            bA, bB = min_c.fA.body, min_c.fB.body
            backup = [(b, b.c0.copy(), b.a0, b.c.copy(), b.a, b.alpha0) for b in (bA, bB)]
            bA.advance(min_alpha)
            bB.advance(min_alpha)
            min_c.update()
            min_c.toi_valid = False
            min_c.toi_count += 1
            if not min_c.touching:
                for b, c0, a0, c, a, al in backup:
                    b.c0, b.a0, b.c, b.a, b.alpha0 = c0, a0, c, a, al
                    b.synchronize_transform()
                continue
            island_bodies, island_contacts = [bA, bB], [min_c]
            for body in (bA, bB):
                if not body.dynamic:
                    continue
                for c in self.contacts:
                    if c is min_c or (c.fA.body is not body and c.fB.body is not body):
                        continue
                    if len(island_contacts) == MAX_TOI_CONTACTS:
                        break
                    other = c.fB.body if c.fA.body is body else c.fA.body
                    if other.dynamic:                                # non-bullets: only static partners join
                        continue
                    c.update()
                    if not c.touching:
                        continue
                    island_contacts.append(c)
                    if other not in island_bodies:
                        island_bodies.append(other)
            h = (f32(1.0) - min_alpha) * f32(dt)
            self._solve_toi_island(island_bodies, island_contacts, h, vel_iters, bA, bB)
            for body in island_bodies:
                if not body.dynamic:
                    continue
                for c in self.contacts:
                    if c.fA.body is body or c.fB.body is body:
                        c.toi_valid = False

    def _solve_toi_island(self, bodies, contacts, h, vel_iters, toiA, toiB):
        """b2Island::SolveTOI"""
        pos = {id(b): [b.c.copy(), b.a] for b in bodies}
        vel = {id(b): [b.v.copy(), b.w] for b in bodies}

        def P(b): return pos[id(b)]
        def Vv(b): return vel[id(b)] if b.dynamic else [V(), f32(0.0)]
        vcs = [VelocityConstraint(c, P, Vv, f32(0.0)) for c in contacts]
        for _ in range(20):                               # subStep.positionIterations
            min_sep = f32(0.0)
            for vc in vcs:
                min_sep = min(min_sep, vc.solve_position(P, baumgarte=TOI_BAUMGARTE, movers=(toiA, toiB)))
            if min_sep >= f32(-1.5) * LINEAR_SLOP:
                break
        for b in (toiA, toiB):                             # leap of faith to the new safe state
            b.c0, b.a0 = pos[id(b)][0].copy(), pos[id(b)][1]
        for vc in vcs:                                     # no warm starting: impulses start at zero
            vc.ni = [f32(0.0)] * len(vc.ni)
            vc.ti = [f32(0.0)] * len(vc.ti)
            vc.initialize(P, Vv)
        for _ in range(vel_iters):
            for vc in vcs:
                vc.solve(Vv)
        for b in bodies:
            if not b.dynamic:
                continue
            c, a = pos[id(b)]
            v, w = vel[id(b)]
            t = h * v
            if dot(t, t) > MAX_TRANSLATION * MAX_TRANSLATION:
                v = (MAX_TRANSLATION / t.length()) * v
            r = h * w
            if r * r > MAX_ROTATION * MAX_ROTATION:
                w = w * (MAX_ROTATION / abs(r))
            c = c + h * v
            a = a + h * w
            b.c, b.a, b.v, b.w = c, a, v, w
            b.synchronize_transform()

    def _solve(self, h, dt_ratio, vel_iters, pos_iters):
        # islands: connected components over touching contacts between dynamic bodies; static bodies do not connect
        parent = {id(b): b for b in self.bodies}

        def find(b):
            while parent[id(b)] is not b:
                b = parent[id(b)]
            return b
        for c in self.contacts:
            if c.touching and c.fA.body.dynamic and c.fB.body.dynamic:
                ra, rb = find(c.fA.body), find(c.fB.body)
                if ra is not rb:
                    parent[id(rb)] = ra
        roots = []
        for b in self.bodies:
            if b.dynamic and find(b) not in roots:
                roots.append(find(b))
        for root in roots:
            bodies = [b for b in self.bodies if b.dynamic and find(b) is root]
            # b2World::Solve: islands grow from awake seeds and wake every body they reach; a component without an awake
            # body is not simulated
            if not any(b.awake for b in bodies):
                continue
            for b in bodies:
                b.set_awake(True)
            contacts = [c for c in self.contacts if c.touching and
                        ((c.fA.body.dynamic and find(c.fA.body) is root) or (c.fB.body.dynamic and find(c.fB.body) is root))]
            self._solve_island(bodies, contacts, h, dt_ratio, vel_iters, pos_iters)

    def _damp(self, h, c):
        if self.damping == 'linear':
            return clamp(f32(1.0) - h * c, 0.0, 1.0)
        return f32(1.0) / (f32(1.0) + h * c)

    def _solve_island(self, bodies, contacts, h, dt_ratio, vel_iters, pos_iters):
        """b2Island::Solve"""
        pos, vel = {}, {}
        for b in bodies:
            b.c0, b.a0 = b.c.copy(), b.a
            v, w = b.v.copy(), b.w
            # (no gravity, no forces)
            v = self._damp(h, b.linear_damping) * v
            w = w * self._damp(h, b.angular_damping)
            pos[id(b)] = [b.c.copy(), b.a]
            vel[id(b)] = [v, w]

        def P(b): return pos[id(b)] if b.dynamic else [b.c, b.a]
        def Vv(b): return vel[id(b)] if b.dynamic else [V(), f32(0.0)]
        vcs = [VelocityConstraint(c, P, Vv, dt_ratio) for c in contacts]
        for vc in vcs:
            vc.initialize(P, Vv)
        for vc in vcs:
            vc.warm_start(Vv)
        for _ in range(vel_iters):
            for vc in vcs:
                vc.solve(Vv)
        for vc in vcs:
            vc.store()
        for b in bodies:
            c, a = pos[id(b)]
            v, w = vel[id(b)]
            t = h * v
            if dot(t, t) > MAX_TRANSLATION * MAX_TRANSLATION:
                v = (MAX_TRANSLATION / t.length()) * v
            r = h * w
            if r * r > MAX_ROTATION * MAX_ROTATION:
                w = w * (MAX_ROTATION / abs(r))
            c = c + h * v
            a = a + h * w
            pos[id(b)] = [c, a]
            vel[id(b)] = [v, w]
        position_solved = False
        for _ in range(pos_iters):
            min_sep = f32(0.0)
            for vc in vcs:
                min_sep = min(min_sep, vc.solve_position(P))
            if min_sep >= f32(-3.0) * LINEAR_SLOP:
                position_solved = True
                break
        for b in bodies:
            b.c, b.a = pos[id(b)]
            b.v, b.w = vel[id(b)]
            b.synchronize_transform()
        if self.allow_sleep:             # b2Island::Solve, the allowSleep block
            min_sleep = f32(3.402823466e+38)
            lin_tol_sqr = LINEAR_SLEEP_TOLERANCE * LINEAR_SLEEP_TOLERANCE
            ang_tol_sqr = ANGULAR_SLEEP_TOLERANCE * ANGULAR_SLEEP_TOLERANCE
            for b in bodies:
                if b.w * b.w > ang_tol_sqr or dot(b.v, b.v) > lin_tol_sqr:
                    b.sleep_time = f32(0.0)
                    min_sleep = f32(0.0)
                else:
                    b.sleep_time = b.sleep_time + h
                    min_sleep = min(min_sleep, b.sleep_time)
            if min_sleep >= TIME_TO_SLEEP and position_solved:
                for b in bodies:
                    b.set_awake(False)


class VelocityConstraint:
    """b2ContactVelocityConstraint + b2ContactPositionConstraint of one contact."""

    def __init__(self, contact, P, Vv, dt_ratio):
        self.c = contact
        self.bA, self.bB = contact.fA.body, contact.fB.body
        self.mA, self.mB = (self.bA.inv_mass if self.bA.dynamic else f32(0.0)), (self.bB.inv_mass if self.bB.dynamic else f32(0.0))
        self.iA, self.iB = (self.bA.inv_I if self.bA.dynamic else f32(0.0)), (self.bB.inv_I if self.bB.dynamic else f32(0.0))
        self.friction, self.restitution = contact.friction, contact.restitution
        self.rA_shape, self.rB_shape = contact.fA.shape.radius, contact.fB.shape.radius
        m = contact.manifold
        self.count = len(m.points)
        self.ni = [dt_ratio * p.normal_impulse for p in m.points]       # warm starting
        self.ti = [dt_ratio * p.tangent_impulse for p in m.points]
        self.lcA, self.lcB = self.bA.local_center, self.bB.local_center

    def _xf(self, b, P):
        c, a = P(b)
        q = Rot(a)
        lc = b.local_center
        return XF(c - rot_mul(q, lc), q)

    def initialize(self, P, Vv):
        """b2ContactSolver::InitializeVelocityConstraints"""
        m = self.c.manifold
        cA, _ = P(self.bA)
        cB, _ = P(self.bB)
        vA, wA = Vv(self.bA)
        vB, wB = Vv(self.bB)
        xfA, xfB = self._xf(self.bA, P), self._xf(self.bB, P)
        normal, points = world_manifold(m, xfA, self.rA_shape, xfB, self.rB_shape)
        self.normal = normal
        self.rA, self.rB, self.nmass, self.tmass, self.bias = [], [], [], [], []
        tangent = cross_vs(normal, 1.0)
        for j in range(self.count):
            rA, rB = points[j] - cA, points[j] - cB
            rnA, rnB = cross(rA, normal), cross(rB, normal)
            kn = self.mA + self.mB + self.iA * rnA * rnA + self.iB * rnB * rnB
            self.nmass.append(f32(1.0) / kn if kn > 0.0 else f32(0.0))
            rtA, rtB = cross(rA, tangent), cross(rB, tangent)
            kt = self.mA + self.mB + self.iA * rtA * rtA + self.iB * rtB * rtB
            self.tmass.append(f32(1.0) / kt if kt > 0.0 else f32(0.0))
            vrel = dot(normal, vB + cross_sv(wB, rB) - vA - cross_sv(wA, rA))
            self.bias.append(-self.restitution * vrel if vrel < -VELOCITY_THRESHOLD else f32(0.0))
            self.rA.append(rA)
            self.rB.append(rB)
        self.block = False
        if self.count == 2:
            rn1A, rn1B = cross(self.rA[0], normal), cross(self.rB[0], normal)
            rn2A, rn2B = cross(self.rA[1], normal), cross(self.rB[1], normal)
            k11 = self.mA + self.mB + self.iA * rn1A * rn1A + self.iB * rn1B * rn1B
            k22 = self.mA + self.mB + self.iA * rn2A * rn2A + self.iB * rn2B * rn2B
            k12 = self.mA + self.mB + self.iA * rn1A * rn2A + self.iB * rn1B * rn2B
            k_max_cond = f32(1000.0)
            if k11 * k11 < k_max_cond * (k11 * k22 - k12 * k12):
                self.K = (k11, k12, k12, k22)
                det = k11 * k22 - k12 * k12
                if det != 0.0:
                    det = f32(1.0) / det
                self.Kinv = (det * k22, -det * k12, -det * k12, det * k11)        # ex.x, ey.x, ex.y, ey.y
                self.block = True
            else:
                self.count = 1          # the constraints are redundant: use one point

    def warm_start(self, Vv):
        vA, vB = Vv(self.bA), Vv(self.bB)
        tangent = cross_vs(self.normal, 1.0)
        for j in range(self.count):
            Pj = self.ni[j] * self.normal + self.ti[j] * tangent
            vA[1] = vA[1] - self.iA * cross(self.rA[j], Pj)
            vA[0] = vA[0] - self.mA * Pj
            vB[1] = vB[1] + self.iB * cross(self.rB[j], Pj)
            vB[0] = vB[0] + self.mB * Pj

    def solve(self, Vv):
        """b2ContactSolver::SolveVelocityConstraints for this contact"""
        sA, sB = Vv(self.bA), Vv(self.bB)
        vA, wA, vB, wB = sA[0], sA[1], sB[0], sB[1]
        normal = self.normal
        tangent = cross_vs(normal, 1.0)
        for j in range(self.count):          # friction first
            dv = vB + cross_sv(wB, self.rB[j]) - vA - cross_sv(wA, self.rA[j])
            vt = dot(dv, tangent)
            lam = self.tmass[j] * (-vt)
            maxf = self.friction * self.ni[j]
            new = clamp(self.ti[j] + lam, -maxf, maxf)
            lam = new - self.ti[j]
            self.ti[j] = new
            Pj = lam * tangent
            vA = vA - self.mA * Pj
            wA = wA - self.iA * cross(self.rA[j], Pj)
            vB = vB + self.mB * Pj
            wB = wB + self.iB * cross(self.rB[j], Pj)
        if self.count == 1 or not self.block:
            for j in range(self.count):
                dv = vB + cross_sv(wB, self.rB[j]) - vA - cross_sv(wA, self.rA[j])
                vn = dot(dv, normal)
                lam = -self.nmass[j] * (vn - self.bias[j])
                new = max(self.ni[j] + lam, f32(0.0))
                lam = new - self.ni[j]
                self.ni[j] = new
                Pj = lam * normal
                vA = vA - self.mA * Pj
                wA = wA - self.iA * cross(self.rA[j], Pj)
                vB = vB + self.mB * Pj
                wB = wB + self.iB * cross(self.rB[j], Pj)
        else:
            a = V(self.ni[0], self.ni[1])
            dv1 = vB + cross_sv(wB, self.rB[0]) - vA - cross_sv(wA, self.rA[0])
            dv2 = vB + cross_sv(wB, self.rB[1]) - vA - cross_sv(wA, self.rA[1])
            vn1, vn2 = dot(dv1, normal), dot(dv2, normal)
            k11, k12, k21, k22 = self.K
            b = V(vn1 - self.bias[0], vn2 - self.bias[1])
            b = b - V(k11 * a.x + k12 * a.y, k21 * a.x + k22 * a.y)

            def apply(x):
                nonlocal vA, wA, vB, wB
                d = x - a
                P1, P2 = d.x * normal, d.y * normal
                vA = vA - self.mA * (P1 + P2)
                wA = wA - self.iA * (cross(self.rA[0], P1) + cross(self.rA[1], P2))
                vB = vB + self.mB * (P1 + P2)
                wB = wB + self.iB * (cross(self.rB[0], P1) + cross(self.rB[1], P2))
                self.ni[0], self.ni[1] = x.x, x.y
            i11, i12, i21, i22 = self.Kinv
            while True:
                # case 1: both points active
                x = V(-(i11 * b.x + i12 * b.y), -(i21 * b.x + i22 * b.y))
                if x.x >= 0.0 and x.y >= 0.0:
                    apply(x)
                    break
                # case 2: point 1 active, point 2 inactive
                x = V(-self.nmass[0] * b.x, 0.0)
                vn2_ = k21 * x.x + b.y
                if x.x >= 0.0 and vn2_ >= 0.0:
                    apply(x)
                    break
                # case 3: point 2 active, point 1 inactive
                x = V(0.0, -self.nmass[1] * b.y)
                vn1_ = k12 * x.y + b.x
                if x.y >= 0.0 and vn1_ >= 0.0:
                    apply(x)
                    break
                # case 4: both inactive
                x = V(0.0, 0.0)
                if b.x >= 0.0 and b.y >= 0.0:
                    apply(x)
                    break
                break        # no solution: give up (as Box2D does)
        sA[0], sA[1], sB[0], sB[1] = vA, wA, vB, wB

    def store(self):
        for j, p in enumerate(self.c.manifold.points[:len(self.ni)]):
            p.normal_impulse, p.tangent_impulse = self.ni[j], self.ti[j]

    def solve_position(self, P, baumgarte=None, movers=None):
        """b2ContactSolver::SolvePositionConstraints (or SolveTOIPositionConstraints with `movers`: only the two TOI bodies
        have mass) for this contact; returns its minimum separation."""
        m = self.c.manifold
        sA, sB = P(self.bA), P(self.bB)
        cA, aA, cB, aB = sA[0], sA[1], sB[0], sB[1]
        min_sep = f32(0.0)
        baum = BAUMGARTE if baumgarte is None else baumgarte
        mA, iA, mB, iB = self.mA, self.iA, self.mB, self.iB
        if movers is not None:
            if not any(self.bA is x for x in movers):
                mA = iA = f32(0.0)
            if not any(self.bB is x for x in movers):
                mB = iB = f32(0.0)
        for j in range(len(m.points)):
            qA, qB = Rot(aA), Rot(aB)
            xfA = XF(cA - rot_mul(qA, self.lcA), qA)
            xfB = XF(cB - rot_mul(qB, self.lcB), qB)
            # b2PositionSolverManifold
            if m.type == 'circles':
                pA, pB = xf_mul(xfA, m.local_point), xf_mul(xfB, m.points[0].local_point)
                normal = (pB - pA).normalized()
                point = f32(0.5) * (pA + pB)
                sep = dot(pB - pA, normal) - self.rA_shape - self.rB_shape
            elif m.type == 'faceA':
                normal = rot_mul(xfA.q, m.local_normal)
                plane = xf_mul(xfA, m.local_point)
                clip = xf_mul(xfB, m.points[j].local_point)
                sep = dot(clip - plane, normal) - self.rA_shape - self.rB_shape
                point = clip
            else:
                normal = rot_mul(xfB.q, m.local_normal)
                plane = xf_mul(xfB, m.local_point)
                clip = xf_mul(xfA, m.points[j].local_point)
                sep = dot(clip - plane, normal) - self.rA_shape - self.rB_shape
                point = clip
                normal = -normal
            rA, rB = point - cA, point - cB
            min_sep = min(min_sep, sep)
            C = clamp(baum * (sep + LINEAR_SLOP), -MAX_LINEAR_CORRECTION, 0.0)
            rnA, rnB = cross(rA, normal), cross(rB, normal)
            K = mA + mB + iA * rnA * rnA + iB * rnB * rnB
            imp = -C / K if K > 0.0 else f32(0.0)
            Pv = imp * normal
            cA = cA - mA * Pv
            aA = aA - iA * cross(rA, Pv)
            cB = cB + mB * Pv
            aB = aB + iB * cross(rB, Pv)
        if self.bA.dynamic:
            sA[0], sA[1] = cA, aA
        if self.bB.dynamic:
            sB[0], sB[1] = cB, aB
        return min_sep


def time_of_impact_edge_circle(E, bA, C, bB):
    """b2TimeOfImpact for a static edge (proxy A, radius = polygon radius) and a circle (proxy B) whose body sweeps linearly
    from (c0, a0) to (c, a) over t in [0, 1].  Between the core shapes (the segment and the circle's centre) b2Distance is
    the point-segment distance; the separation function is the e_faceA one on the edge normal (two simplex vertices on the
    edge, one on the circle), as long as the closest point lies inside the segment.  Returns (state, t)."""
    total = E.radius + C.radius
    target = max(LINEAR_SLOP, total - f32(3.0) * LINEAR_SLOP)
    tol = f32(0.25) * LINEAR_SLOP
    v1, v2 = xf_mul(bA.xf, E.v1), xf_mul(bA.xf, E.v2)
    e = v2 - v1

    def centre(t):
        t = f32(t)
        c = (f32(1.0) - t) * bB.c0 + t * bB.c
        a = (f32(1.0) - t) * bB.a0 + t * bB.a
        q = Rot(a)
        xf = XF(c - rot_mul(q, bB.local_center), q)
        return xf_mul(xf, C.p)

    def dist_and_axis(t):
        p = centre(t)
        u = dot(e, p - v1) / dot(e, e)
        u = clamp(u, 0.0, 1.0)
        closest = v1 + u * e
        d = p - closest
        return d.length(), p
    t1 = f32(0.0)
    for _ in range(20):
        dist, p = dist_and_axis(t1)
        if dist <= 0.0:
            return 'overlapped', f32(0.0)
        if dist < target + tol:
            return 'touching', t1
        # separation function (e_faceA): normal of the edge pointing at the circle at t1
        n = cross_vs(e, 1.0).normalized()
        mid = f32(0.5) * (v1 + v2)
        if dot(p - mid, n) < 0.0:
            n = -n

        def sep(t):
            return dot(centre(t) - mid, n)
        done = False
        t2 = f32(1.0)
        for _push in range(8):
            s2 = sep(t2)
            if s2 > target + tol:
                return 'separated', f32(1.0)
            if s2 > target - tol:
                t1 = t2
                break
            s1 = sep(t1)
            if s1 < target - tol:
                return 'failed', t1
            if s1 <= target + tol:
                return 'touching', t1
            a1, a2 = t1, t2
            for it in range(50):
                if it & 1:
                    t = a1 + (target - s1) * (a2 - a1) / (s2 - s1)
                else:
                    t = f32(0.5) * (a1 + a2)
                s = sep(t)
                if abs(s - target) < tol:
                    t2 = t
                    break
                if s > target:
                    a1, s1 = t, s
                else:
                    a2, s2 = t, s
        else:
            done = True
        if done:
            break
    return 'failed', t1


def world_manifold(m, xfA, rA, xfB, rB):
    """b2WorldManifold::Initialize: (normal, points)"""
    if m.type == 'circles':
        normal = V(1.0, 0.0)
        pA, pB = xf_mul(xfA, m.local_point), xf_mul(xfB, m.points[0].local_point)
        if dist_sq(pA, pB) > EPSILON * EPSILON:
            normal = (pB - pA).normalized()
        cA, cB = pA + rA * normal, pB - rB * normal
        return normal, [f32(0.5) * (cA + cB)]
    if m.type == 'faceA':
        normal = rot_mul(xfA.q, m.local_normal)
        plane = xf_mul(xfA, m.local_point)
        pts = []
        for p in m.points:
            clip = xf_mul(xfB, p.local_point)
            cA = clip + (rA - dot(clip - plane, normal)) * normal
            cB = clip - rB * normal
            pts.append(f32(0.5) * (cA + cB))
        return normal, pts
    normal = rot_mul(xfB.q, m.local_normal)
    plane = xf_mul(xfB, m.local_point)
    pts = []
    for p in m.points:
        clip = xf_mul(xfA, p.local_point)
        cB = clip + (rB - dot(clip - plane, normal)) * normal
        cA = clip - rA * normal
        pts.append(f32(0.5) * (cA + cB))
    return -normal, pts
