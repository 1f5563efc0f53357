"""Debug helper: one scene of tests/test_parity_fuzz_gpu.py (SEED=...), stepped like the test; prints every field that differs
after each step.  KB_HIP_LIB selects an experiment build."""
import os, sys
import numpy as np, torch
sys.path.insert(0, '.')
import tests.test_parity_fuzz_gpu as F
from tests.test_parity_gpu import cpu, OBJ_FIELDS

seed = int(os.environ.get('SEED', '0'))
real_assert = F.assert_same
def loud(osim, gsim, what='', fields=()):
    torch.cuda.synchronize()
    bad = False
    for f in tuple(fields) + ('ws_cnt',):
        a, b = getattr(osim, f), cpu(getattr(gsim, f))
        b = b.reshape(a.shape)
        if not np.array_equal(a, b, equal_nan=False):
            idx = np.argwhere(~(a == b))
            print(what, f, 'differs at', idx[:8].tolist(), 'oracle', a[tuple(idx[0])], 'gpu', b[tuple(idx[0])])
            bad = True
    if bad:
        print('oracle status', osim.status, 'gpu status', cpu(gsim.status))
        np.set_printoptions(precision=5, suppress=True, linewidth=200)
        for f in ('x', 'y', 'ox', 'oy', 'otheta', 'ovx', 'ovy', 'ow'):
            if getattr(osim, f, None) is not None and getattr(gsim, f, None) is not None:
                print(f, 'oracle', np.asarray(getattr(osim, f)).ravel()[:16]); print(f, 'gpu   ', cpu(getattr(gsim, f)).ravel()[:16])
        raise SystemExit(1)
F.assert_same = loud
if os.environ.get('SINGLE'):
    # split every fused step into single-substep launches and compare after each one
    real_make = F.make_pair
    def make_pair(*a, **k):
        osim, gsim = real_make(*a, **k)
        ostep, gstep = osim.step, gsim.step
        state = {'n': 0}
        def o_step(n=1, **kw):
            state['pending'] = (n, kw)
        def g_step(n=1, actions=None, light_action=None, **kw):
            on, okw = state['pending']
            for i in range(n):
                ostep(1, **okw)
                gstep(1, actions=actions, light_action=light_action, **kw)
                state['n'] += 1
                loud(osim, gsim, 'substep %d' % state['n'], ('x', 'y', 'theta') + (('sleep_time',) if osim.cfg.allow_sleep else ()) + (tuple(OBJ_FIELDS[3:]) if osim.cfg.num_objects else ()))
                if os.environ.get('WS'):
                    from tests.test_parity_gpu import assert_ws_same
                    assert_ws_same(osim, gsim, 'substep %d' % state['n'])
        osim.step, gsim.step = o_step, g_step
        return osim, gsim
    F.make_pair = make_pair
F.test_random_scene.__wrapped__(seed) if hasattr(F.test_random_scene, '__wrapped__') else F.test_random_scene(seed)
print('seed', seed, 'ok')
