"""Oracle checks for pushable objects with Box2D's full contact model: boxes / convex polygons,
friction, rotation, two-point manifolds.  The solver is `parity unpinned` (Box2D is absent), so these are
known-answer values of the restated formulas and physical invariants, not reference vectors."""
import numpy as np
import pytest

from oracle import oracle as O

W = 25.0            # world units per metre
SLOP = 0.005
POLY_R = 0.01
KL = 1.0 / (1.0 + 0.1 * 0.8)


def _sim(num_bots=1, shapes=('box',), sizes=None, **kw):
    M = len(shapes)
    sizes = sizes or [(0.15, 0.15)] * M
    obj_shape, obj_verts, obj_radius, obj_nverts = [], [], [], []
    for sh, sz in zip(shapes, sizes):
        if sh == 'box':
            obj_shape.append(O.SHAPE_BOX); obj_verts.append([[sz[0] / 2 * W, sz[1] / 2 * W]]); obj_radius.append(0.0); obj_nverts.append(4)
        elif sh == 'circle':
            obj_shape.append(O.SHAPE_CIRCLE); obj_verts.append([[0, 0]]); obj_radius.append(sz[0]); obj_nverts.append(0)
        else:                                   # explicit counter-clockwise polygon
            obj_shape.append(O.SHAPE_POLYGON); obj_verts.append([(x * W, y * W) for x, y in sz]); obj_radius.append(0.0); obj_nverts.append(len(sz))
    cfg = O.default_config(1, num_bots, O.DRIVE_VELOCITY, num_objects=M, obj_shape=obj_shape, obj_verts=obj_verts,
                           obj_radius=obj_radius, obj_nverts=obj_nverts, **kw)
    return O.OracleSim(cfg)


def _park_bots(sim, where=(0.9, 0.65)):
    n = sim.cfg.num_bots
    xy = np.array([[where[0] - 0.05 * i, where[1]] for i in range(n)])
    sim.set_poses_m(xy[None], np.zeros((1, n)))
    sim.set_actions(None)


def test_box_mass_and_inertia_known_answer():
    # a 0.15 m quad of density 2 has mass 28.125 and inertia 65.918 (world units)
    sim = _sim()
    _park_bots(sim)
    sim.set_objects_m([[[0.0, 0.0]]])
    sim.ovx[...] = 1.0                       # free flight: v' = v / (1 + h c), x' = x + h v'
    sim.ow[...] = 0.5
    sim.step(1)
    assert np.isclose(sim.ovx[0, 0], KL, rtol=1e-6)
    assert np.isclose(sim.ox[0, 0], 0.1 * KL, rtol=1e-6)
    assert np.isclose(sim.otheta[0, 0], 0.1 * 0.5 * KL, rtol=1e-6)
    # one kilobot (mass 1.06912) hits the box face centrally at relative speed v: plastic impact shares the
    # momentum, v_box = m_bot v / (m_bot + m_box) (restitution 0, warm start empty, single contact)
    sim2 = _sim()
    gap = 0.075 + 0.0165 + 0.0002
    sim2.set_poses_m([[[-gap, 0.0]]], [[0.0]])
    sim2.set_objects_m([[[0.0, 0.0]]])
    sim2.set_actions(np.array([[[0.01, 0.0]]], np.float32))
    sim2.step(1)
    m_bot, m_box = 2.0 * np.pi * (0.0165 * W) ** 2, 28.125
    v = 0.01 * W * KL
    assert sim2.count_contacts(0, True)[2] == 1
    assert np.isclose(sim2.ovx[0, 0], m_bot * v / (m_bot + m_box), rtol=2e-5)
    assert abs(sim2.ow[0, 0]) < 1e-7 and abs(sim2.ovy[0, 0]) < 1e-7


def test_off_centre_push_spins_the_box_with_the_right_sign():
    for y0, sign in ((0.03, -1.0), (-0.03, 1.0)):
        sim = _sim()
        sim.set_poses_m([[[-0.12, y0]]], [[0.0]])
        sim.set_objects_m([[[0.0, 0.0]]])
        sim.set_actions(np.array([[[0.01, 0.0]]], np.float32))
        sim.step(100)
        assert sim.status[0] == 0
        assert sim.ox[0, 0] > 0.0
        assert np.sign(sim.otheta[0, 0]) == sign and abs(sim.otheta[0, 0]) > 1e-3


def test_impulse_of_a_lever_arm_contact_known_answer():
    # kilobot hits the face at lever arm r: lambda = v / (1/m_bot + 1/m_box + r^2 / I_box)
    sim = _sim()
    y0 = 0.04
    gap = 0.075 + 0.0165 + 0.0002
    sim.set_poses_m([[[-gap, y0]]], [[0.0]])
    sim.set_objects_m([[[0.0, 0.0]]])
    sim.set_actions(np.array([[[0.01, 0.0]]], np.float32))
    sim.step(1)
    m_bot, m_box, i_box = 2.0 * np.pi * (0.0165 * W) ** 2, 28.125, 28.125 * (3.75 ** 2 + 3.75 ** 2) / 12.0
    r = y0 * W
    v = 0.01 * W * KL
    lam = v / (1.0 / m_bot + 1.0 / m_box + r * r / i_box)
    assert np.isclose(sim.ovx[0, 0], lam / m_box, rtol=5e-5)
    assert np.isclose(sim.ow[0, 0], -r * lam / i_box, rtol=5e-5)


def test_box_rests_flat_against_a_wall_two_point_manifold():
    sim = _sim(num_bots=2)
    # two kilobots push the box symmetrically into the right wall
    sim.set_poses_m([[[0.75, 0.03], [0.75, -0.03]]], [[0.0, 0.0]])
    sim.set_objects_m([[[0.85, 0.0]]], [[0.05]])          # slightly rotated: must settle flat
    act = np.array([[[0.01, 0.0], [0.01, 0.0]]], np.float32)
    sim.set_actions(act)
    sim.step(400)
    assert sim.status[0] == 0
    # face at x = 1.0 - gap: separation of the touching face within [-3 slop, 2 polygon radii]
    face = sim.ox[0, 0] / W + 0.075
    assert 1.0 - 2 * POLY_R / W - 1e-6 <= face + 2 * POLY_R / W + 1e-3
    assert face <= 1.0 + 3 * SLOP / W
    assert abs(sim.otheta[0, 0]) < 5e-3                     # lying flat on the wall
    ows = sim.ows_acc[0, 0, 8 + 2]                          # warm-start entry (object 0, right wall)
    assert ows[0] >= 0 and ows[3] >= 0, "two manifold points expected"
    assert ows[1] > 0 and ows[4] > 0                        # both carry normal impulse
    # at rest: no creeping
    x0 = sim.ox.copy()
    sim.step(50)
    assert abs(sim.ox[0, 0] - x0[0, 0]) < 2e-3


def test_box_box_collision_conserves_linear_momentum():
    sim = _sim(shapes=('box', 'box'), sizes=[(0.15, 0.15), (0.2, 0.1)])
    _park_bots(sim)
    sim.set_objects_m([[[-0.09, 0.01], [0.09, -0.02]]], [[0.1, -0.3]])
    sim.ovx[0] = [1.2, -0.7]
    sim.ovy[0] = [0.1, 0.05]
    m = np.array([2.0 * (0.15 * W) ** 2, 2.0 * 0.2 * W * 0.1 * W])
    touched = False
    for _ in range(12):
        p0 = np.array([np.dot(m, sim.ovx[0]), np.dot(m, sim.ovy[0])])
        sim.step(1)
        p1 = np.array([np.dot(m, sim.ovx[0]), np.dot(m, sim.ovy[0])])
        assert np.allclose(p1, KL * p0, rtol=1e-4, atol=1e-4)          # contacts are internal forces
        touched |= sim.count_contacts(0, True)[2] > 0
    assert touched
    assert sim.ovx[0, 0] < 1.2 * KL ** 12 - 0.05                        # ... and they did exchange momentum


def test_disc_box_and_disc_disc_contacts():
    sim = _sim(shapes=('circle', 'box', 'circle'), sizes=[(0.05, 0), (0.15, 0.15), (0.04, 0)])
    _park_bots(sim)
    sim.set_objects_m([[[-0.2, 0.0], [0.0, 0.0], [-0.4, 0.01]]])
    sim.ovx[0] = [0.0, 0.0, 12.0]
    for _ in range(200):
        sim.step(1)
    assert sim.status[0] == 0
    # the chain disc -> disc -> box was set in motion, nothing interpenetrates beyond the solver's slop
    assert sim.ox[0, 1] > 0.0
    d01 = np.hypot(sim.ox[0, 0] - sim.ox[0, 2], sim.oy[0, 0] - sim.oy[0, 2])
    assert d01 > (0.05 + 0.04) * W - 4 * SLOP
    assert (sim.ox[0, 1] - 0.075 * W) - sim.ox[0, 0] > 0.05 * W - 4 * SLOP - 0.2


def test_friction_spins_a_disc_sliding_along_a_wall():
    mu = np.sqrt(0.01 * 0.2)
    sim = _sim(shapes=('circle',), sizes=[(0.05, 0)])
    _park_bots(sim)
    # disc pressed on the bottom wall by its own motion (vy < 0) while sliding in +x: friction torque is clockwise
    sim.set_objects_m([[[0.0, -0.75 + 0.05 + 0.5 * POLY_R / W]]])
    sim.ovx[...] = 1.0
    sim.ovy[...] = -1.0
    sim.step(1)
    m, r = 2.0 * np.pi * (0.05 * W) ** 2, 0.05 * W
    jn = m * KL                          # normal impulse that stops vy
    assert np.isclose(sim.ovy[0, 0], 0.0, atol=1e-6)
    assert np.isclose(sim.ovx[0, 0], KL - mu * jn / m, rtol=1e-4)
    assert sim.ow[0, 0] < 0.0
    # lever arm = distance of the contact point (midway between the two skins) from the centre
    arm = abs(sim.ow[0, 0]) * (0.5 * m * r * r) / (mu * jn)
    assert abs(arm - r) < 2 * POLY_R
    # without friction: no spin
    sim0 = _sim(shapes=('circle',), sizes=[(0.05, 0)], obj_friction=0.0)
    _park_bots(sim0)
    sim0.set_objects_m([[[0.0, -0.75 + 0.05 + 0.5 * POLY_R / W]]])
    sim0.ovx[...] = 1.0
    sim0.ovy[...] = -1.0
    sim0.step(1)
    assert sim0.ow[0, 0] == 0.0 and np.isclose(sim0.ovx[0, 0], KL, rtol=1e-6)


def test_triangle_polygon_object():
    # the reference's Triangle (body.py:265-275) scaled to 0.15 x 0.15, recentred on its centroid, hull order of
    # b2PolygonShape::Set (lowest of the right-most points first, counter-clockwise)
    raw = np.array([(0.0, 0.0), (0.0, 1.0), (-0.5, 0.0)]) / np.array([0.5, 1.0]) * 0.15
    tri = raw - raw.mean(0)
    sim = _sim(shapes=('poly',), sizes=[[tuple(v) for v in tri]])
    sim.set_poses_m([[[-0.2, 0.0]]], [[0.0]])
    sim.set_objects_m([[[0.0, 0.0]]])
    sim.set_actions(np.array([[[0.01, 0.0]]], np.float32))
    sim.step(200)
    assert sim.status[0] == 0
    assert sim.ox[0, 0] > 0.05                      # pushed along
    assert abs(sim.otheta[0, 0]) > 1e-3             # and turned: the push line misses the centroid
    # free flight known answer: mass = density * area = 2 * (0.5 * 3.75 * 3.75)
    sim2 = _sim(shapes=('poly',), sizes=[[tuple(v) for v in tri]])
    gap = 0.05 + 0.0165 + 0.0002                    # left face of the recentred triangle is at x = -0.1 ... no: hit the apex side
    sim2.set_poses_m([[[0.9, 0.6]]], [[0.0]])
    sim2.set_objects_m([[[0.0, 0.0]]])
    sim2.ovx[...] = 1.0
    sim2.step(1)
    assert np.isclose(sim2.ovx[0, 0], KL, rtol=1e-6)
    del gap


def test_rotated_box_hits_a_wall_corner_first_and_stays_inside():
    sim = _sim()
    _park_bots(sim)
    sim.set_objects_m([[[0.8, 0.0]]], [[0.6]])
    sim.ovx[...] = 1.5
    worst = 0.0
    for _ in range(150):
        sim.step(1)
        th = float(sim.otheta[0, 0])
        c, s = np.cos(th), np.sin(th)
        corners = np.array([[sx * 1.875, sy * 1.875] for sx in (-1, 1) for sy in (-1, 1)])
        wx = sim.ox[0, 0] + c * corners[:, 0] - s * corners[:, 1]
        worst = max(worst, wx.max() - 25.0)
    assert sim.status[0] == 0
    assert worst < 0.1                                # never deeper than a few slops into the wall
    assert abs(sim.ovx[0, 0]) < 1e-2


def test_threads_do_not_change_results():
    def run(threads):
        cfg = O.default_config(6, 8, O.DRIVE_VELOCITY, num_objects=2, obj_shape=[O.SHAPE_BOX, O.SHAPE_CIRCLE],
                               obj_verts=[[[0.075 * W, 0.05 * W]], [[0, 0]]], obj_radius=[0.0, 0.06])
        sim = O.OracleSim(cfg)
        rng = np.random.default_rng(5)
        sim.set_poses_m(rng.uniform(-0.25, 0.25, (6, 8, 2)), rng.uniform(-3, 3, (6, 8)))
        sim.set_objects_m(np.tile(np.array([[0.0, 0.0], [0.12, 0.05]]), (6, 1, 1)), rng.uniform(-1, 1, (6, 2)))
        sim.set_actions(rng.uniform([0, -1.5], [0.01, 1.5], (6, 8, 2)).astype(np.float32))
        sim.step(40, threads=threads)
        return sim.poses_m(), sim.objects_m()
    a, b = run(1), run(4)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def _lform_fixtures(width=0.15, height=0.15):
    """The reference's LForm (body.py:277-287) as lib.body.Polygon scales and recentres it: two boxes."""
    raw = np.array([[(-0.05, 0.0), (0.1, 0.0), (0.1, 0.3), (-0.05, 0.3)],
                    [(0.1, 0.0), (0.1, -0.15), (-0.2, -0.15), (-0.2, 0.0)]])
    size = raw.max((0, 1)) - raw.min((0, 1))
    v = raw / size * np.array((width, height))
    cen, area = np.zeros(2), 0.0
    for vs in v:
        a = 0.5 * abs(np.dot(vs[:, 0], np.roll(vs[:, 1], 1)) - np.dot(vs[:, 1], np.roll(vs[:, 0], 1)))
        area += a
        cen += vs.mean(0) * a
    return v - cen / area


def _ccw_from_lowest_right(vs):
    """b2PolygonShape::Set order for a convex quad given in any rotation direction."""
    vs = np.asarray(vs)
    x, y = vs[:, 0], vs[:, 1]
    if np.dot(x, np.roll(y, -1)) - np.dot(y, np.roll(x, -1)) < 0:
        vs = vs[::-1]
    i0 = max(range(len(vs)), key=lambda i: (vs[i, 0], -vs[i, 1]))
    return np.roll(vs, -i0, axis=0)


def test_two_fixture_body_mass_centre_and_push():
    fx = [_ccw_from_lowest_right(f) for f in _lform_fixtures()]
    cfg = O.default_config(1, 1, O.DRIVE_VELOCITY, num_objects=1, num_fixtures=2, obj_fixture_body=[0, 0],
                           obj_shape=[O.SHAPE_POLYGON, O.SHAPE_POLYGON], obj_nverts=[4, 4],
                           obj_verts=[[(x * W, y * W) for x, y in f] for f in fx], obj_radius=[0.0, 0.0])
    sim = O.OracleSim(cfg)
    sim.set_poses_m([[[0.9, 0.6]]], [[0.0]])
    sim.set_actions(None)
    sim.set_objects_m([[[0.0, 0.0]]], [[0.4]])
    sim.ovx[...] = 1.0
    sim.ow[...] = 0.0
    sim.step(1)
    # free flight: the body ORIGIN is what the state holds (Body.get_pose); without spin it moves with the centre of mass
    assert np.isclose(sim.ovx[0, 0], KL, rtol=1e-6) and np.isclose(sim.ox[0, 0], 0.1 * KL, rtol=1e-5, atol=1e-7)
    assert np.isclose(sim.otheta[0, 0], 0.4, rtol=1e-6)
    # pure spin about the centre of mass: the origin (which the reference recentres on the area-weighted vertex mean,
    # equal to the centre of mass for two rectangles) stays put up to rounding
    sim.set_objects_m([[[0.0, 0.0]]], [[0.0]])
    sim.ovx[...] = 0.0
    sim.ow[...] = 1.0
    sim.step(5)
    assert abs(sim.ox[0, 0]) < 1e-4 and abs(sim.oy[0, 0]) < 1e-4 and sim.otheta[0, 0] > 0.3
    # a kilobot pushes the long arm: the L turns and moves, both fixtures can be touched
    sim2 = O.OracleSim(cfg)
    sim2.set_poses_m([[[-0.2, 0.06]]], [[0.0]])
    sim2.set_objects_m([[[0.0, 0.0]]], [[0.0]])
    sim2.set_actions(np.array([[[0.01, 0.0]]], np.float32))
    sim2.step(250)
    assert sim2.status[0] == 0 and sim2.ox[0, 0] > 0.01 and abs(sim2.otheta[0, 0]) > 1e-3


def test_two_fixture_body_against_wall_and_box():
    fx = [_ccw_from_lowest_right(f) for f in _lform_fixtures()]
    cfg = O.default_config(1, 1, O.DRIVE_VELOCITY, num_objects=2, num_fixtures=3, obj_fixture_body=[0, 0, 1],
                           obj_shape=[O.SHAPE_POLYGON, O.SHAPE_POLYGON, O.SHAPE_BOX], obj_nverts=[4, 4, 4],
                           obj_verts=[[(x * W, y * W) for x, y in f] for f in fx] + [[(0.05 * W, 0.05 * W)]],
                           obj_radius=[0.0, 0.0, 0.0])
    sim = O.OracleSim(cfg)
    sim.set_poses_m([[[-0.9, 0.6]]], [[0.0]])
    sim.set_actions(None)
    sim.set_objects_m([[[0.55, 0.0], [0.8, 0.02]]], [[0.3, 0.1]])
    sim.ovx[0] = [10.0, 0.0]
    m = None
    for k in range(120):
        sim.step(1)
    assert sim.status[0] == 0
    # the L shoved the box to the right wall; nothing ended up outside the arena
    assert sim.ox[0, 1] / W > 0.85 and sim.ox[0, 1] / W < 1.0 - 0.05 + 0.002
    assert (sim.ows_acc[0, :3, :, 0] >= 0).any()          # manifolds between fixtures / walls were active
    del m


def _vertex_margin(sim, hw, hh):
    """smallest wall distance (world units) over the vertices of box object 0"""
    x, y, a = sim.ox[0, 0], sim.oy[0, 0], sim.otheta[0, 0]
    v = np.array([[-hw, -hh], [hw, -hh], [hw, hh], [-hw, hh]]) * W
    wx = x + np.cos(a) * v[:, 0] - np.sin(a) * v[:, 1]
    wy = y + np.sin(a) * v[:, 0] + np.cos(a) * v[:, 1]
    return min((1.0 * W - np.abs(wx)).min(), (0.75 * W - np.abs(wy)).min())


@pytest.mark.parametrize('angle,spin', [(0.3, 3.0), (0.0, 0.0), (np.pi / 4, -5.0), (2.0, 0.5)])
def test_continuous_step_stops_a_thrown_box_at_the_wall(angle, spin):
    """b2World::SolveTOI for a polygon body: a box flying at 1.2 units per substep crosses the 0.015 penetration the
    discrete solver tolerates, so a TOI event puts it back at the target separation (core polygon 0.005 from the wall
    line, tolerance 0.00125); without the continuous step its corner ends up far beyond the wall."""
    margins = {}
    for toi in (0, 1):
        sim = _sim(toi_walls=toi)
        _park_bots(sim, where=(-0.5, 0.0))
        sim.set_objects_m([[[0.8, 0.1]]], [[angle]])
        sim.ovx[...] = 12.0
        sim.ovy[...] = 1.0
        sim.ow[...] = spin
        worst = 1e9
        for _ in range(8):
            sim.step(1)
            worst = min(worst, _vertex_margin(sim, 0.075, 0.075))
        margins[toi] = worst
        assert sim.status[0] == 0
    assert margins[1] > 0.005 - 0.00125 - 1e-4
    assert margins[0] < -0.05
    # the TOI sub-solve removed the approach velocity (the remaining time is integrated with the solved velocity)
    assert sim.ovx[0, 0] < 1.0


def test_continuous_step_of_a_box_has_wall_friction():
    """The TOI contacts are ordinary Box2D contacts: sliding along the wall during the sub-solve is braked by friction
    sqrt(f_obj f_wall), so the tangential velocity after the impact is lower than free flight alone would leave it."""
    out = {}
    for mu in (0.0, 0.8):
        sim = _sim(toi_walls=1, wall_friction=mu, obj_friction=0.5)
        _park_bots(sim, where=(-0.5, 0.0))
        sim.set_objects_m([[[0.8, 0.0]]], [[0.0]])
        sim.ovx[...] = 12.0
        sim.ovy[...] = 4.0
        sim.step(5)
        out[mu] = float(sim.ovy[0, 0])
        assert sim.ovx[0, 0] < 0.5                            # it did hit the wall
    assert np.isclose(out[0.0], 4.0 * KL ** 5, rtol=1e-4)
    assert out[0.8] < out[0.0] - 0.5
