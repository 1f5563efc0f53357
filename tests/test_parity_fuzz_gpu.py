"""Randomised parity sweep: many small random scenes (bot counts, drive laws, light models, object shapes with random
fixture-to-body maps, spawn densities, solver paths), each compared bit for bit with the oracle substep by substep."""
import os

import numpy as np
import pytest
import torch

from oracle import oracle as O
from tests import scenes
from tests.test_parity_gpu import make_pair, assert_same, assert_ws_same, cpu, dev, OBJ_FIELDS, TRIANGLE

pytestmark = pytest.mark.gpu

# how many scenes of a sweep leave at the capacity early-return (the overflow region must be measurable): counted
# per process, written behind the last scene to gpurun_out/fuzz_summary.txt (and printed with `pytest -s`)
COUNTS = {'scenes': 0, 'capacity_flag': 0, 'device_limit_flag': 0}


@pytest.fixture(scope='module', autouse=True)
def _fuzz_summary():
    yield
    line = ('fuzz sweep: %(scenes)d scenes, %(capacity_flag)d left at the contact-capacity flag (status bit 0, oracle and device agree on it), '
            '%(device_limit_flag)d at a device-only staging limit (status bit 2)' % COUNTS)
    print(line)
    try:
        os.makedirs('gpurun_out', exist_ok=True)
        with open(os.path.join('gpurun_out', 'fuzz_summary.txt'), 'a') as f:
            f.write(line + '\n')
    except OSError:
        pass


def _random_objects(rng):
    """Random bodies: circles, boxes, triangles and two-/three-part compounds; fixtures shuffled across bodies."""
    from gym_kilobots_amd.lib.body import _hull_order
    nobj = int(rng.integers(1, 5))
    fixtures = []                      # (shape, verts or None, radius, body)
    for b in range(nobj):
        kind = rng.choice(['circle', 'box', 'tri', 'two', 'three']) if len(fixtures) <= 5 else rng.choice(['circle', 'box'])
        if kind == 'circle':
            fixtures.append((O.SHAPE_CIRCLE, None, float(rng.uniform(0.03, 0.07)), b))
        elif kind == 'box':
            fixtures.append((O.SHAPE_BOX, [[float(rng.uniform(0.03, 0.09)) * 25.0, float(rng.uniform(0.03, 0.09)) * 25.0]], 0.0, b))
        elif kind == 'tri':
            fixtures.append((O.SHAPE_POLYGON, [[x * 25.0, y * 25.0] for x, y in TRIANGLE], 0.0, b))
        else:
            parts = 2 if kind == 'two' else 3
            w, h = float(rng.uniform(0.04, 0.07)), float(rng.uniform(0.02, 0.04))
            for k in range(parts):     # a row of boxes glued together (as explicit quads)
                cx = (k - (parts - 1) / 2) * 2 * w
                quad = [(cx - w, -h), (cx + w, -h), (cx + w, h * (1 + 0.5 * k)), (cx - w, h)]
                fixtures.append((O.SHAPE_POLYGON, [list(q) for q in _hull_order([(x * 25.0, y * 25.0) for x, y in quad])], 0.0, b))
    fixtures = fixtures[:8]
    nobj = len({f[3] for f in fixtures})
    remap = {b: i for i, b in enumerate(sorted({f[3] for f in fixtures}))}
    order = rng.permutation(len(fixtures))
    fixtures = [fixtures[i] for i in order]
    # a circle must be alone on its body: guaranteed by construction
    pad = 8 - len(fixtures)
    kw = dict(num_objects=nobj, num_fixtures=len(fixtures), obj_fixture_body=[remap[f[3]] for f in fixtures] + [0] * pad,
              obj_shape=[f[0] for f in fixtures], obj_nverts=[0 if f[1] is None else len(f[1]) for f in fixtures],
              obj_radius=[f[2] for f in fixtures], obj_verts=[[[0.0, 0.0]] if f[1] is None else f[1] for f in fixtures])
    return nobj, kw


@pytest.mark.parametrize('seed', list(range(int(os.environ.get('KB_FUZZ_SEEDS', '40')))))
def test_random_scene(seed):
    rng = np.random.default_rng(1000 + seed)
    N = int(rng.choice([1, 2, 7, 16, 33, 64, 100, 128, 200, 256, 300]))
    if os.environ.get('KB_FUZZ_BIG'):
        N = int(rng.choice([512, 700, 1000, 1024, 1024]))
    E = int(rng.integers(1, 5))
    mode = int(rng.choice([O.DRIVE_VELOCITY, O.DRIVE_VELOCITY, O.DRIVE_ACCEL, O.DRIVE_MOTORS, O.DRIVE_SIMPLE_PHOTOTAXIS, O.DRIVE_PHOTOTAXIS]))
    light = O.LIGHT_NONE
    if mode in (O.DRIVE_SIMPLE_PHOTOTAXIS, O.DRIVE_PHOTOTAXIS) or rng.random() < 0.3:
        light = int(rng.choice([O.LIGHT_CIRCULAR, O.LIGHT_GRADIENT, O.LIGHT_MOMENTUM]))
    kw = dict(solver_mode=int(rng.choice([0, 0, 0, 1, 2, 3, 4])), toi_walls=int(rng.random() < 0.8))
    # arena size (grid dimensions, LDS footprint, workgroups per CU) and iteration counts: from their own stream, so that
    # the scenes of the seeds keep their layout
    rng2 = np.random.default_rng(777000 + seed)
    W, H = 2.0, 1.5
    if rng2.random() < 0.5:
        W, H = [(1.0, 0.8), (1.2, 0.9), (3.0, 2.0), (0.6, 0.6), (2.0, 0.5)][int(rng2.integers(0, 5))]
        kw.update(world_width=W, world_height=H)
    if rng2.random() < 0.3:
        kw.update(vel_iters=int(rng2.choice([6, 3, 1])), pos_iters=int(rng2.choice([4, 2, 1, 0])))
    # sleeping (b2World doSleep): in 40 % of the scenes, with phases in which kilobots rest so that islands fall asleep and
    # are woken again (drawn last from the second stream: the older scenes keep their layout)
    sleeping = rng2.random() < 0.4
    if sleeping:
        kw.update(allow_sleep=1)
    # mixed drive laws (KB_DRIVE_MIXED): in a quarter of the scenes, every kilobot its own law and the
    # classes' densities (drawn after everything else from the second stream)
    mixed = rng2.random() < 0.25
    if mixed:
        mode = O.DRIVE_MIXED
        kw.update(mode_density=[2.0, 2.0, 1.0, 1.0, 1.0])
    sx, sy = W / 2.0, H / 1.5
    with_objects = rng.random() < 0.6
    nobj = 0
    if with_objects:
        nobj, okw = _random_objects(rng)
        kw.update(okw)
    sigma = float(rng.choice([0.03, 0.08, 0.2, 0.5]))
    if os.environ.get('KB_FUZZ_BIG'):
        sigma = float(rng.choice([0.12, 0.2, 0.3, 0.5]))
    xy = np.clip(rng.normal(scale=sigma, size=(E, N, 2)) + rng.uniform(-0.5, 0.5, (E, 1, 2)) * [sx, sy],
                 [-W / 2 + 0.03, -H / 2 + 0.03], [W / 2 - 0.03, H / 2 - 0.03])
    th = rng.uniform(-np.pi, np.pi, (E, N))
    osim, gsim = make_pair(E, N, mode, light, xy=xy, th=th, **kw)
    if nobj:
        objs = rng.uniform([-0.8, -0.55], [0.8, 0.55], (E, nobj, 2)) * [sx, sy]
        oth = rng.uniform(-np.pi, np.pi, (E, nobj))
        osim.set_objects_m(objs, oth)
        gsim.set_objects_m(objs, oth)
        v0 = rng.uniform(-6, 6, (E, nobj)).astype(np.float32)
        osim.ovx[...] = v0
        gsim.ovx.copy_(dev(v0))
    if light != O.LIGHT_NONE:
        lx = (rng.uniform(-0.5, 0.5, osim.light_x.shape) * min(sx, sy)).astype(np.float32)
        osim.light_x[...] = lx
        gsim.light_x.copy_(dev(lx))
    if mixed:
        laws = rng2.integers(0, 5, (E, N)).astype(np.uint8)
        osim.bot_mode[...] = laws
        gsim.bot_mode.copy_(dev(laws))
        ml, mr = rng2.integers(0, 256, (E, N)).astype(np.uint8), rng2.integers(0, 256, (E, N)).astype(np.uint8)
        ml[rng2.random((E, N)) < 0.3] = 0
        for name, v in (('motor_l', ml), ('motor_r', mr)):
            getattr(osim, name)[...] = v
            getattr(gsim, name).copy_(dev(v))
    fields = ('x', 'y', 'theta') + (OBJ_FIELDS[3:] if nobj else ())
    if sleeping:
        fields += ('sleep_time',) + (('osleep',) if nobj else ())
        if mode in (O.DRIVE_MOTORS, O.DRIVE_PHOTOTAXIS):      # a third of the kilobots has its motors off
            off = rng2.random((E, N)) < 0.35
            for name in ('motor_l', 'motor_r'):
                v = getattr(osim, name)
                v[off] = 0
                getattr(gsim, name).copy_(dev(v))
    la_dim = {O.LIGHT_NONE: 0, O.LIGHT_GRADIENT: 1}.get(light, 2)
    for k in range(12):
        la = None if la_dim == 0 or k % 3 == 2 else rng.uniform(-0.02, 0.02, (E, la_dim)).astype(np.float32)
        n_sub = int(rng.choice([1, 1, 3, 10]))
        if mode in (O.DRIVE_VELOCITY, O.DRIVE_ACCEL, O.DRIVE_MIXED):
            a = scenes.random_actions(E, N, seed=5000 + 31 * seed + k)
            if sleeping:      # resting phases: everybody for a few steps, then a random half
                if k % 6 in (1, 2, 3):
                    a[...] = 0.0
                elif k % 6 == 4:
                    a[rng2.random((E, N)) < 0.5] = 0.0
                if mode == O.DRIVE_ACCEL and k % 6 in (1, 2, 3):
                    a[..., 0] = -0.005        # decelerate to a halt (the command is clamped at 0)
                    a[..., 1] = 0.0
            osim.set_actions(a)
            osim.step(n_sub, light_action=la)
            gsim.step(n_sub, actions=dev(a), light_action=None if la is None else dev(la))
        else:
            osim.step(n_sub, light_action=la)
            gsim.step(n_sub, light_action=None if la is None else dev(la))
        torch.cuda.synchronize()
        so, sg = osim.status, cpu(gsim.status)
        if k == 0:
            COUNTS['scenes'] += 1
        if ((so | sg) & 1).any() or (sg & 4).any():
            COUNTS['capacity_flag' if ((so | sg) & 1).any() else 'device_limit_flag'] += 1
            # contact capacity exceeded (absurdly dense spawn) or more kilobots on one fixture / in one rank group than
            # the device stages: which contacts are dropped is unspecified, only the flag is
            assert ((so & 1) == (sg & 1)).all(), 'capacity flag differs: %s vs %s' % (so, sg)
            return
        assert_same(osim, gsim, 'seed %d (N=%d E=%d mode=%d light=%d objects=%d %s) step %d' % (seed, N, E, mode, light, nobj, kw.get('solver_mode'), k), fields)
        assert_ws_same(osim, gsim, 'seed %d step %d' % (seed, k))
    assert np.array_equal(osim.status & 3, cpu(gsim.status) & 3)     # (bits 2, 3 are limits of the device staging only)
