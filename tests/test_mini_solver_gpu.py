"""The HIP path itself (through the C ABI) against the INDEPENDENT restatement of Box2D's step (tools/box2d_mini.py,
tests/golden/mini_solver.json): the same fixtures that pin the oracle on the CPU, replayed on the device with the same
float32 tolerances -- discrete solver, friction, block solver, polygon / circle / wall manifolds, continuous step."""
import numpy as np
import pytest
import torch

from oracle import oracle as O
from tests.test_oracle_vs_mini_solver import FIX

pytestmark = pytest.mark.gpu


def _sim_for(sc):
    from gym_kilobots_amd.sim import KilobotSim
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
    kb = np.array(sc['kilobots'], np.float64)
    mixed = kb.shape[1] > 5
    if mixed:
        kw.update(mode_density=[2.0, 2.0, 1.0, 1.0, 1.0])
    g = KilobotSim(1, len(kb), O.DRIVE_MIXED if mixed else O.DRIVE_VELOCITY, **kw)
    dev = g.x.device
    if mixed:
        g.bot_mode.copy_(torch.tensor(np.where(kb[None, :, 5] == 2.0, O.DRIVE_VELOCITY, O.DRIVE_MOTORS), dtype=torch.uint8, device=dev))
        g.motor_l.zero_()
        g.motor_r.zero_()
    g.x.copy_(torch.tensor(kb[None, :, 0], dtype=torch.float32, device=dev))
    g.y.copy_(torch.tensor(kb[None, :, 1], dtype=torch.float32, device=dev))
    g.theta.copy_(torch.tensor(kb[None, :, 2], dtype=torch.float32, device=dev))
    g.forget_contacts()
    g.set_actions(torch.tensor(kb[None, :, 3:5], dtype=torch.float32, device=dev).contiguous())
    for name, key in (('ox', 'x'), ('oy', 'y'), ('otheta', 'theta'), ('ovx', 'vx'), ('ovy', 'vy'), ('ow', 'w')):
        if objs:
            getattr(g, name).copy_(torch.tensor([[o[key] for o in objs]], dtype=torch.float32, device=dev))
    return g


@pytest.mark.parametrize('name', sorted(FIX))
def test_hip_path_follows_the_independent_solver(name):
    sc = FIX[name]
    g = _sim_for(sc)
    tol = sc['tol']
    for k, ref in enumerate(sc['trajectory']):
        for first, cmds in sc.get('commands', []):
            if first == k:
                g.set_actions(torch.tensor([cmds], dtype=torch.float32, device=g.x.device).contiguous())
        g.step(1)
        if sc.get('sleep'):          # the same substep falls asleep / wakes on the device as in the independent derivation
            st = g.sleep_time[0].double().cpu().numpy()
            want_s = np.array(ref['sleep'])
            assert np.array_equal(st < 0, want_s < 0) and np.abs(st - want_s).max() <= 1e-6, (name, k, 'sleep time', st, want_s)
            if sc['objects']:
                so = g.osleep[0].double().cpu().numpy()
                assert np.array_equal(so < 0, np.array(ref['osleep']) < 0), (name, k, 'object asleep', so)
        got = torch.stack([g.x[0], g.y[0], g.theta[0]], -1).double().cpu().numpy()
        want = np.array(ref['kilobots'])
        assert np.abs(got[:, :2] - want[:, :2]).max() <= tol, (name, k, 'kilobot position')
        assert np.abs(got[:, 2] - want[:, 2]).max() <= 10 * tol, (name, k, 'kilobot angle')
        if sc['objects']:
            o = torch.stack([g.ox[0], g.oy[0], g.otheta[0], g.ovx[0], g.ovy[0], g.ow[0]], -1).double().cpu().numpy()
            w = np.array(ref['objects'])
            assert np.abs(o[:, :2] - w[:, :2]).max() <= tol, (name, k, 'object position')
            assert np.abs(o[:, 2] - w[:, 2]).max() <= 10 * tol, (name, k, 'object angle')
            assert np.abs(o[:, 3:] - w[:, 3:]).max() <= 200 * tol, (name, k, 'object velocity')
    assert int(g.status.max().item()) == 0
