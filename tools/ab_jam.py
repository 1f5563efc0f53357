#!/usr/bin/env python3
"""A/B timing of experiment libraries on the jammed swarm (1024 envs x 1024 SimplePhototaxis kilobots around a light, one giant
island; -DKB_ONLY_JAM builds):  python tools/ab_jam.py name1 name2 ...   (each in a fresh process)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def one():
    import numpy as np
    import torch
    from gym_kilobots_amd.sim import KilobotSim
    from gym_kilobots_amd import _native as nat
    from tests import scenes
    E, N = int(os.environ.get('KB_JAM_ENVS', '1024')), 1024
    sim = KilobotSim(E, N, nat.DRIVE_SIMPLE_PHOTOTAXIS, nat.LIGHT_CIRCULAR, light_radius=2.0, ws_slots=8, allow_sleep=0)
    xy1, th1 = scenes.lattice_spawn(8, N, seed=1000)
    sim.set_poses_m(np.tile(xy1, (E // 8, 1, 1)), np.tile(th1, (E // 8, 1)))
    sim.light_x.zero_(); sim.light_y.zero_()
    for _ in range(350):
        sim.step(1)
    torch.cuda.synchronize()
    res = []
    for rep in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            sim.step(1)
        b.record()
        torch.cuda.synchronize()
        res.append(a.elapsed_time(b) / 20)
    print('%s: %d envs, %s ms per launch, %.3e kilobot-steps/s, %.1f contacts per env, status %d, envs/CU %d' % (os.environ.get('KB_HIP_LIB', 'product').split('_')[-1], E, ' '.join('%.4f' % r for r in res), E * N / (min(res) * 1e-3),
          float(sim.ws_cnt.sum(dtype=torch.int64).item()) / E, int(sim.status.max().item()), sim.resident_envs_per_cu), flush=True)


if __name__ == '__main__':
    if len(sys.argv) == 1:
        one()
    else:
        for name in sys.argv[1:]:
            lib = os.path.join(ROOT, 'gym_kilobots_amd', 'libkilobots_hip_%s.so' % name) if name != 'product' else os.path.join(ROOT, 'gym_kilobots_amd', 'libkilobots_hip.so')
            subprocess.call([sys.executable, os.path.abspath(__file__)], env=dict(os.environ, KB_HIP_LIB=lib))
