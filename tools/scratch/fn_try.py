import sys, numpy as np, torch
sys.path.insert(0, '.')
from gym_kilobots_amd.sim import KilobotSim
from tests import scenes
E, N = 4, 1024
sim = KilobotSim(E, N)
xy, th = scenes.lattice_spawn(E, N, seed=1)
sim.set_poses_m(xy, th)
a = torch.from_numpy(scenes.random_actions(E, N, seed=2)).cuda()
print('launch', flush=True)
sim.step(1, actions=a)
torch.cuda.synchronize()
print('done', sim.status.cpu().numpy(), flush=True)
