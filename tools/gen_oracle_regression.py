#!/usr/bin/env python3
"""Writes tests/golden/oracle_regression.npz: end states of a few small scenes computed by the CPU oracle
(oracle/kb_oracle.c).  This is NOT reference data (the reference's Box2D is absent, DESIGN.md "Oracle"): it pins the
oracle's OWN semantics, so that a change of the specification (contact order, solver arithmetic, shapes ...) is a
visible, deliberate act -- re-run this script and commit the new file together with the change."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
from tests import scenes  # noqa: E402


def scenes_list():
    W = 25.0
    tri = [[1.25, -1.25], [1.25, 2.5], [-2.5, -1.25]]
    out = []
    # 1: crowd, velocity control
    E, N = 2, 96
    xy, th = scenes.gaussian_spawn(E, N, sigma=0.12, seed=11)
    out.append(('crowd', O.default_config(E, N), xy, th, None, None, 40))
    # 2: walls + continuous step, motors
    E, N = 1, 48
    xy, th = scenes.gaussian_spawn(E, N, sigma=0.3, seed=12)
    xy[:, :, 0] = np.clip(xy[:, :, 0] + 0.7, -0.97, 0.97)
    out.append(('walls', O.default_config(E, N, O.DRIVE_MOTORS), xy, th, None, None, 60))
    # 3: objects of every kind: box, disc, triangle, two-fixture body
    E, N = 2, 64
    xy, th = scenes.gaussian_spawn(E, N, sigma=0.2, seed=13)
    kw = dict(num_objects=4, num_fixtures=5, obj_fixture_body=[0, 1, 2, 3, 3, 0, 0, 0],
              obj_shape=[O.SHAPE_BOX, O.SHAPE_CIRCLE, O.SHAPE_POLYGON, O.SHAPE_POLYGON, O.SHAPE_POLYGON],
              obj_nverts=[4, 0, 3, 4, 4], obj_radius=[0.0, 0.05, 0.0, 0.0, 0.0],
              obj_verts=[[[0.075 * W, 0.05 * W]], [[0, 0]], tri,
                         [[1.0, -0.5], [1.0, 0.5], [-1.0, 0.5], [-1.0, -0.5]], [[3.0, -0.5], [3.0, 1.0], [1.0, 0.5], [1.0, -0.5]]])
    objs = np.tile(np.array([[0.2, 0.1], [-0.2, 0.15], [0.0, -0.25], [-0.3, -0.2]])[None], (E, 1, 1))
    oth = np.tile(np.array([0.3, 0.0, -0.8, 1.1])[None], (E, 1))
    out.append(('objects', O.default_config(E, N, **kw), xy, scenes.toward_objects_theta(xy), objs, oth, 50))
    return out


def run(cfg, xy, th, objs, oth, steps):
    sim = O.OracleSim(cfg)
    sim.set_poses_m(xy, th)
    if objs is not None:
        sim.set_objects_m(objs, oth)
        sim.ovx[...] = 2.0
    E, N = cfg.num_envs, cfg.num_bots
    for k in range(steps):
        if cfg.drive_mode == O.DRIVE_VELOCITY:
            sim.set_actions(scenes.random_actions(E, N, seed=300 + k))
        sim.step(1)
    res = {'x': sim.x.copy(), 'y': sim.y.copy(), 'theta': sim.theta.copy()}
    if objs is not None:
        res.update(ox=sim.ox.copy(), oy=sim.oy.copy(), otheta=sim.otheta.copy(), ovx=sim.ovx.copy(), ow=sim.ow.copy())
    return res


def compute():
    out = {}
    for name, cfg, xy, th, objs, oth, steps in scenes_list():
        for k, v in run(cfg, xy, th, objs, oth, steps).items():
            out['%s/%s' % (name, k)] = v
    return out


if __name__ == '__main__':
    path = os.path.join(ROOT, 'tests', 'golden', 'oracle_regression.npz')
    np.savez_compressed(path, **compute())
    print('wrote', path, os.path.getsize(path), 'bytes')
