import os, sys
sys.path.insert(0, '.')
import numpy as np, torch
from gym_kilobots_amd.sim import KilobotSim
from gym_kilobots_amd import _native as nat
from tests import scenes
E, N = 768, 1024
sim = KilobotSim(E, N, nat.DRIVE_SIMPLE_PHOTOTAXIS, nat.LIGHT_CIRCULAR, light_radius=2.0, ws_slots=8, allow_sleep=0)
xy1, th1 = scenes.lattice_spawn(8, N, seed=1000)
sim.set_poses_m(np.tile(xy1, (E // 8, 1, 1)), np.tile(th1, (E // 8, 1)))
sim.light_x.zero_(); sim.light_y.zero_()
out = []
done = 0
for target in (60, 120, 180, 240, 350):
    while done < target:
        sim.step(1); done += 1
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): sim.step(1)
    b.record(); torch.cuda.synchronize(); done += 10
    out.append('%d: %.3f ms (%.0f contacts)' % (target, a.elapsed_time(b) / 10, float(sim.ws_cnt.sum(dtype=torch.int64).item()) / E))
print(os.environ.get('KB_HIP_LIB', 'product').split('_')[-1], ' | '.join(out), 'status', int(sim.status.max().item()))
