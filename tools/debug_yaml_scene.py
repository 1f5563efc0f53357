"""Debug helper: the YAML box-pushing scene of tests/test_env_api_gpu.py stepped one substep at a time on the HIP path and
on the oracle-backed env; prints the first field that differs.  Knobs: TOI, MODE (solver_mode), POS / VEL (iterations),
KB_HIP_LIB (experiment build)."""
import os, sys, numpy as np, torch, yaml
sys.path.insert(0, '.')
from tests.test_env_api_gpu import YAML_BOXES
from tests.oracle_backend import OracleBackend
from gym_kilobots_amd.envs import YamlKilobotsEnv
from gym_kilobots_amd.sim import KilobotSim
conf = yaml.load(YAML_BOXES, Loader=yaml.Loader)
extra = dict(toi_walls=int(os.environ.get('TOI', '1')), solver_mode=int(os.environ.get('MODE', '0')), pos_iters=int(os.environ.get('POS', '10')), vel_iters=int(os.environ.get('VEL', '10')))
def gf(*a, **k):
    k.update(extra); return KilobotSim(*a, **k)
def of(*a, **k):
    k.update(extra); return OracleBackend(*a, **k)
g = YamlKilobotsEnv(configuration=conf, sim_factory=gf)
o = YamlKilobotsEnv(configuration=conf, sim_factory=of)
np.random.seed(11); g.reset(); np.random.seed(11); o.reset()
gs, os_ = g.sim, o.sim
F = ('x','y','theta','ox','oy','otheta','ovx','ovy','ow')
def cmp(tag):
    bad = False
    for f in F:
        a = getattr(gs,f).cpu().numpy(); b = getattr(os_,f).numpy().reshape(a.shape)
        if not np.array_equal(a,b):
            d = np.abs(a.astype(np.float64)-b); print(tag, f, 'max', d.max(), 'at', np.unravel_index(d.argmax(), d.shape), 'n', (d>0).sum()); bad = True
    a = gs.ows_acc.cpu().numpy(); b = os_.o.ows_acc.reshape(a.shape)
    if not np.array_equal(a,b):
        idx = np.argwhere(a!=b); print(tag,'ows differs at', idx[:6].tolist(), a[tuple(idx[0])], b[tuple(idx[0])]); bad = True
    return bad
cmp('after reset')
la = torch.tensor([[0.01, 0.0]], dtype=torch.float32)
for k in range(10):
    gs.step(1, light_action=la.cuda()); os_.step(1, light_action=la)
    bad = cmp('substep %d' % k)
    try:
        from tests.test_parity_gpu import assert_ws_same
        assert_ws_same(os_.o, gs, 'substep %d' % k)
    except AssertionError as ex:
        print('WS', ex); bad = True
    if bad:
        break
print('status', gs.status.cpu().numpy(), os_.status.numpy())
