#!/usr/bin/env python3
"""Launch time of the step while a swarm clusters: 1024 SimplePhototaxis kilobots per env drive towards a light in the middle
of the arena and jam into ONE island (the regime of the pushing tasks at swarm scale; the headline workload keeps ~3 contacts
per kilobot in many small islands).  Prints contacts per env and ms per 1-substep launch every 50 substeps.
    python tools/cluster_probe.py [envs=1024] [bots=1024] [substeps=600]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gym_kilobots_amd.sim import KilobotSim
from gym_kilobots_amd import _native as nat
from tests import scenes

E = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
S = int(sys.argv[3]) if len(sys.argv) > 3 else 600
sim = KilobotSim(E, N, nat.DRIVE_SIMPLE_PHOTOTAXIS, nat.LIGHT_CIRCULAR, light_radius=2.0, ws_slots=8,
                 allow_sleep=int(os.environ.get('KB_SLEEP', '0')))
xy1, th1 = scenes.lattice_spawn(8, N, seed=1000)
reps = (E + 7) // 8
sim.set_poses_m(np.tile(xy1, (reps, 1, 1))[:E], np.tile(th1, (reps, 1))[:E])
sim.light_x.zero_(); sim.light_y.zero_()
print('lds %d B per env, %d threads, %d envs per CU' % (sim.lds_bytes, sim.block_threads, sim.resident_envs_per_cu))
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
for k0 in range(0, S, 50):
    ev[0].record()
    for k in range(50):
        sim.step(1)
    ev[1].record()
    torch.cuda.synchronize()
    c = sim.ws_cnt.sum(dim=1, dtype=torch.int64).float()
    r = torch.sqrt(sim.x ** 2 + sim.y ** 2).mean() / 25.0
    print('substeps %4d..%4d: %.3f ms per launch, %.3e kilobot-steps/s | contacts/env mean %.0f max %.0f | mean distance to the light %.3f m | status 0x%x'
          % (k0, k0 + 49, ev[0].elapsed_time(ev[1]) / 50, E * N / (ev[0].elapsed_time(ev[1]) / 50 * 1e-3), c.mean(), c.max(), float(r), sim.status_bits()))
