"""Analytic and invariant checks of the oracle's contact solver (Box2D semantics restated;
SURVEY.md section 4 items 3).  CPU only."""
import numpy as np

from oracle import oracle as O
from tests import scenes

R = 0.0165
SLOP_M = 0.005 / 25


def test_two_bots_head_on_stop_at_contact_distance():
    sim = O.OracleSim(O.default_config(1, 2))
    sim.set_poses_m(np.array([[[-0.03, 0.0], [0.03, 0.0]]]), np.array([[0.0, np.pi]]))
    sim.set_actions(np.array([[[0.01, 0.0], [0.01, 0.0]]], np.float32))
    for _ in range(60):
        sim.step(1)
    p = sim.poses_m()[0]
    d = p[1, 0] - p[0, 0]
    # resting distance: 2r - slop (position solver pushes until separation >= -slop), symmetric, no y motion
    assert abs(d - (2 * R - SLOP_M)) < 0.6 * SLOP_M
    assert abs(p[0, 0] + p[1, 0]) < 1e-6 and abs(p[0, 1]) < 1e-7 and abs(p[1, 1]) < 1e-7
    assert sim.count_contacts(0) == (1, 0)
    # frictionless contacts never turn a kilobot
    assert sim.theta[0, 0] == 0.0 and abs(sim.theta[0, 1] - np.float32(np.pi)) == 0.0


def test_bot_pushes_into_wall_and_rests():
    sim = O.OracleSim(O.default_config(1, 1))
    sim.set_poses_m(np.array([[[0.95, 0.1]]]), np.array([[0.0]]))
    sim.set_actions(np.array([[[0.01, 0.0]]], np.float32))
    for _ in range(80):
        sim.step(1)
    p = sim.poses_m()[0, 0]
    # centre rests at r + polygonRadius - slop from the wall line (edge skin 0.01 world units)
    rest = 1.0 - (R + 0.01 / 25 - SLOP_M)
    assert abs(p[0] - rest) < 0.6 * SLOP_M
    assert abs(p[1] - 0.1) < 1e-7
    assert sim.count_contacts(0) == (0, 1)


def test_oblique_wall_contact_slides_without_friction():
    sim = O.OracleSim(O.default_config(1, 1))
    sim.set_poses_m(np.array([[[0.0, 0.72]]]), np.array([[np.pi / 4]]))
    sim.set_actions(np.array([[[0.01, 0.0]]], np.float32))
    xs = []
    for _ in range(100):
        sim.step(1)
        xs.append(sim.poses_m()[0, 0, 0])
    # tangential speed along the top wall stays v cos(45deg) * damping factor
    v_t = (xs[-1] - xs[-11]) / 1.0
    assert abs(v_t - 0.01 * np.cos(np.pi / 4) / 1.08) < 2e-5


def test_dense_scene_invariants():
    E, N = 4, 128
    xy, th = scenes.gaussian_spawn(E, N, sigma=0.12, seed=5)
    sim = O.OracleSim(O.default_config(E, N))
    sim.set_poses_m(xy, th)
    for k in range(10):
        sim.set_actions(scenes.random_actions(E, N, seed=50 + k))
        sim.step(10)
    assert int(sim.status.max()) & 1 == 0
    p = sim.poses_m()
    assert np.isfinite(p).all()
    for e in range(E):
        d = np.hypot(p[e, :, None, 0] - p[e, None, :, 0], p[e, :, None, 1] - p[e, None, :, 1])
        d[np.diag_indices(N)] = 1.0
        # after the initial overlaps are resolved nobody is deeper than slop + one substep of approach
        assert d.min() > 2 * R - 3 * SLOP_M - 0.0051
    assert (np.abs(p[..., 0]) < 1.0).all() and (np.abs(p[..., 1]) < 0.75).all()


def test_env_permutation_and_thread_count_do_not_matter():
    E, N = 6, 64
    xy, th = scenes.gaussian_spawn(E, N, sigma=0.1, seed=6)
    a = scenes.random_actions(E, N, seed=7)
    s1 = O.OracleSim(O.default_config(E, N))
    s1.set_poses_m(xy, th)
    s1.set_actions(a)
    s1.step(20, threads=1)
    perm = np.array([3, 0, 5, 1, 4, 2])
    s2 = O.OracleSim(O.default_config(E, N))
    s2.set_poses_m(xy[perm], th[perm])
    s2.set_actions(a[perm])
    s2.step(10, threads=4)
    s2.step(10, threads=2)
    assert np.array_equal(s1.x[perm], s2.x) and np.array_equal(s1.y[perm], s2.y) and np.array_equal(s1.theta[perm], s2.theta)


# ---- pushable circular objects (BASELINE config 4) ------------------------------------------------
def test_bot_pushes_object_head_on():
    sim = O.OracleSim(O.default_config(1, 1, num_objects=1))
    sim.set_poses_m(np.array([[[-0.12, 0.0]]]), np.array([[0.0]]))
    sim.set_objects_m(np.array([[[0.0, 0.0]]]))
    sim.set_actions(np.array([[[0.01, 0.0]]], np.float32))
    for _ in range(200):
        sim.step(1)
    bot, obj = sim.poses_m()[0, 0], sim.objects_m()[0, 0]
    assert obj[0] > 0.02 and abs(obj[1]) < 1e-7 and obj[2] == 0.0          # pushed along +x, never spun
    assert abs((obj[0] - bot[0]) - (R + 0.075 - SLOP_M)) < 0.6 * SLOP_M     # resting contact distance
    # momentum bookkeeping of the steady state: the object moves slower than a free kilobot (0.01 / 1.08)
    v_obj = sim.ovx[0, 0] / 25
    assert 0.001 < v_obj < 0.01 / 1.08
    assert sim.count_contacts(0, True) == (0, 0, 1)


def test_object_is_stopped_by_the_wall_and_by_another_object():
    sim = O.OracleSim(O.default_config(1, 6, num_objects=2))
    xy = np.array([[[0.55, 0.02 * (i - 2.5)] for i in range(6)]])
    sim.set_poses_m(xy, np.zeros((1, 6)))
    sim.set_objects_m(np.array([[[0.66, 0.0], [0.83, 0.0]]]))
    sim.set_actions(np.tile([0.01, 0.0], (1, 6, 1)).astype(np.float32))
    for _ in range(400):
        sim.step(1)
    o = sim.objects_m()[0]
    # the far object rests against the right wall (x = 1 - r - skin + slop), the near one against it
    assert abs(o[1, 0] - (1.0 - 0.075 - 0.01 / 25 + SLOP_M)) < 1.2 * SLOP_M
    assert abs((o[1, 0] - o[0, 0]) - (0.15 - SLOP_M)) < 1.2 * SLOP_M
    nb, nw, no = sim.count_contacts(0, True)
    assert no >= 2          # object-object and object-wall (the bots slide off the disc eventually)
    assert int(sim.status.max()) == 0
