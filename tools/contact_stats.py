#!/usr/bin/env python3
"""Distribution over envs of the bench workload's settled state: contacts per env, kilobots near the walls (TOI candidates)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from gym_kilobots_amd.sim import KilobotSim
E, N = 4096, 1024
dev = torch.device('cuda:0')
OBJ = int(os.environ.get('KB_STATS_OBJECTS', '0'))          # cfg4: KB_STATS_OBJECTS=4 [KB_STATS_BOXES=1]
okw = {}
if OBJ and os.environ.get('KB_STATS_BOXES') == '1':
    okw = dict(obj_shape=[1] * OBJ, obj_nverts=[4] * OBJ, obj_verts=[[[0.075 * 25.0, 0.075 * 25.0]]] * OBJ)
sim = KilobotSim(E, N, device=dev, num_objects=OBJ, **okw)
x, y, th, acts = bench.make_scene(torch, E, N, dev, 0, 0, OBJ)
sim.x.copy_(x); sim.y.copy_(y); sim.theta.copy_(th); sim.forget_contacts()
if OBJ:
    sim.set_objects_m(np.tile(bench.CFG4_OBJECTS[None, :OBJ], (E, 1, 1)))
cap_l = sim.lds_staging_entries
print('lds %d B per env, %d contacts staged in LDS (an env with more takes the global staging slice)' % (sim.lds_bytes, cap_l))
for k in range(int(sys.argv[1]) if len(sys.argv) > 1 else 150):
    sim.step(1, actions=acts[k % 8])
    if k % 50 == 49 or k in (99, 119, 149):
        c = sim.ws_cnt.sum(dim=1, dtype=torch.int64).float()
        total = 0.4125 + 0.01
        m = torch.minimum(torch.minimum(sim.x + 25.0, 25.0 - sim.x), torch.minimum(sim.y + 18.75, 18.75 - sim.y))
        near = (m <= total + 0.03).sum(dim=1).float()
        q = lambda t, p: float(torch.quantile(t, p))
        print('envs over the LDS staging: %d of %d' % (int((c > cap_l).sum()), E))
        print('substep %d contacts/env mean %.1f std %.1f p50 %.0f p99 %.0f p99.9 %.0f max %.0f | near-wall bots mean %.1f p99 %.0f max %.0f | status %d'
              % (k + 1, c.mean(), c.std(), q(c, .5), q(c, .99), q(c, .999), c.max(), near.mean(), q(near, .99), near.max(), int(sim.status.max())))
