"""Row a10 / f1 (the Box2D solver): the oracle against an INDEPENDENT second derivation of Box2D's discrete step.

oracle/kb_oracle.c and the HIP kernel are twins of one restatement, so their bit-exact agreement cannot reveal a shared
transcription error of Box2D.  tools/box2d_mini.py is a separate float32 restatement written in Box2D's own structure
(bodies, fixtures, persistent contacts in creation order, b2ContactSolver per contact); tests/golden/mini_solver.json holds
its trajectories for small scenes (circle-circle, circle-wall, polygon-circle with lever arm, box-wall with the two-point
block solver and friction, disc-disc with friction, box-box, and -- scenes `*__toi` -- the continuous step of kilobots and
discs against the walls: b2TimeOfImpact, Advance, the TOI sub-solve; without it the oracle is 150 .. 18 000 tolerances away).  The oracle must follow them within a float32 tolerance
(stated per scene in the fixture: 2e-5 .. 2e-4 world units = 1 .. 8 micrometres for single-contact scenes, where the
sweep order cannot matter; 2e-3 for scenes with several contacts, where Box2D's creation order and the oracle's canonical
order legitimately differ).  box2d-py itself cannot be installed here: row a10 stays "parity unpinned" (DESIGN.md)."""
import json
import os

import numpy as np
import pytest

from oracle import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIX = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'mini_solver.json')))['scenes']


def _oracle_for(sc):
    objs = sc['objects']
    kw = dict(toi_walls=1 if sc.get('toi') else 0, damping_model=1 if sc['damping'] == 'linear' else 0,
              allow_sleep=1 if sc.get('sleep') else 0)
    if objs:
        pad = O.MAX_OBJECTS - len(objs)
        kw.update(num_objects=len(objs),
                  obj_shape=[O.SHAPE_CIRCLE if o['shape'] == 'circle' else O.SHAPE_BOX for o in objs] + [0] * pad,
                  obj_nverts=[0 if o['shape'] == 'circle' else 4 for o in objs] + [0] * pad,
                  obj_radius=[(o['r'] / 25.0) if o['shape'] == 'circle' else 0.0 for o in objs] + [0.075] * pad,
                  obj_verts=[[[0.0, 0.0]] if o['shape'] == 'circle' else [[o['hx'], o['hy']]] for o in objs] + [[[0.0, 0.0]]] * pad)
    N = len(sc['kilobots'])
    kb = np.array(sc['kilobots'], np.float64)
    mixed = kb.shape[1] > 5                      # a density column: kilobots of different classes (KB_DRIVE_MIXED)
    if mixed:
        kw.update(mode_density=[2.0, 2.0, 1.0, 1.0, 1.0])
    o = O.OracleSim(O.default_config(1, N, O.DRIVE_MIXED if mixed else O.DRIVE_VELOCITY, **kw))
    if mixed:                                    # density 2: velocity control; density 1: the motor law with both motors off
        o.bot_mode[0] = np.where(kb[:, 5] == 2.0, O.DRIVE_VELOCITY, O.DRIVE_MOTORS)
        o.motor_l[...] = 0
        o.motor_r[...] = 0
    o.x[0], o.y[0], o.theta[0] = kb[:, 0].astype(np.float32), kb[:, 1].astype(np.float32), kb[:, 2].astype(np.float32)
    o.set_actions(kb[None, :, 3:5].astype(np.float32))
    for m, ob in enumerate(objs):
        o.ox[0, m], o.oy[0, m], o.otheta[0, m] = ob['x'], ob['y'], ob['theta']
        o.ovx[0, m], o.ovy[0, m], o.ow[0, m] = ob['vx'], ob['vy'], ob['w']
    return o


@pytest.mark.parametrize('name', sorted(FIX))
def test_oracle_follows_the_independent_solver(name):
    sc = FIX[name]
    o = _oracle_for(sc)
    tol = sc['tol']
    touched = False
    slept = woke = False
    for k, ref in enumerate(sc['trajectory']):
        for first, cmds in sc.get('commands', []):            # the scene's commands change at these steps
            if first == k:
                o.set_actions(np.array(cmds, np.float32)[None])
        o.step(1)
        if sc.get('sleep'):
            # b2Body::m_sleepTime of every body, -1 = asleep: the same substep falls asleep / wakes in both derivations
            for got_s, want_s in ((o.sleep_time[0], ref['sleep']), (o.osleep[0][:len(sc['objects'])], ref['osleep'])):
                want_s = np.array(want_s, np.float64)
                assert np.array_equal(np.asarray(got_s) < 0, want_s < 0), (name, k, 'asleep flags', got_s, want_s)
                assert np.abs(np.asarray(got_s, np.float64) - want_s).max(initial=0.0) <= 1e-6, (name, k, 'sleep time')
            now = np.array(ref['sleep'] + ref['osleep']) < 0
            slept |= bool(now.any())
            woke |= slept and not now.all() and k > 0 and bool((np.array(sc['trajectory'][k - 1]['sleep'] + sc['trajectory'][k - 1]['osleep']) < 0)[~now].any())
        got = np.stack([o.x[0], o.y[0], o.theta[0]], -1).astype(np.float64)
        want = np.array(ref['kilobots'])
        assert np.abs(got[:, :2] - want[:, :2]).max() <= tol, (name, k, 'kilobot position', got, want)
        assert np.abs(got[:, 2] - want[:, 2]).max() <= 10 * tol, (name, k, 'kilobot angle')
        if sc['objects']:
            g = np.stack([o.ox[0], o.oy[0], o.otheta[0], o.ovx[0], o.ovy[0], o.ow[0]], -1).astype(np.float64)
            w = np.array(ref['objects'])
            assert np.abs(g[:, :2] - w[:, :2]).max() <= tol, (name, k, 'object position', g, w)
            assert np.abs(g[:, 2] - w[:, 2]).max() <= 10 * tol, (name, k, 'object angle', g, w)
            assert np.abs(g[:, 3:] - w[:, 3:]).max() <= 200 * tol, (name, k, 'object velocity', g, w)
        nb, nw, no = o.count_contacts(0, True)
        touched |= (nb + nw + no) > 0
    assert touched or sc.get('contact_free'), 'the scene never made a contact'
    if sc.get('sleep'):
        assert slept, 'nothing fell asleep'
        if 'wake' in name or 'woken' in name:
            assert woke, 'nothing woke up'
    assert int(o.status.max()) == 0


def test_fixture_is_what_the_mini_solver_produces():
    """The committed fixture equals a fresh run of tools/box2d_mini.py (two scenes, bit for bit through JSON)."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import gen_mini_solver_golden as G
    for name in ('head_on', 'box_hits_wall'):
        traj, touched = G.run(G.SCENES[name])
        assert touched
        assert traj == FIX[name]['trajectory']
