#!/usr/bin/env python3
"""A/B timing of experiment libraries on the cfg3 workload (benchmark instantiation only, e.g. -DKB_ONLY_BENCH builds):
    python tools/ab_bench.py name1 name2 ...      # gym_kilobots_amd/libkilobots_hip_<name>.so, each in a fresh process
Prints ms per launch (K back-to-back launches between one event pair, settled scene as in bench.py)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def one():
    import torch
    import bench
    from gym_kilobots_amd.sim import KilobotSim
    dev = torch.device('cuda:0')
    E, N = 4096, 1024
    sim = KilobotSim(E, N, device=dev, allow_sleep=int(os.environ.get('KB_AB_SLEEP', '0')))
    x, y, th, acts = bench.make_scene(torch, E, N, dev, 0, 0, 0)
    sim.x.copy_(x); sim.y.copy_(y); sim.theta.copy_(th)
    sim.forget_contacts()
    k = 0
    for _ in range(125):
        sim.step(1, actions=acts[k % 8]); k += 1
    torch.cuda.synchronize()
    res = []
    for rep in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(40):
            sim.step(1, actions=acts[k % 8]); k += 1
        b.record()
        torch.cuda.synchronize()
        res.append(a.elapsed_time(b) / 40)
    print('%s: %s ms per launch, %.1f contacts per env, status %d' % (os.environ.get('KB_HIP_LIB', 'product').split('_')[-1], ' '.join('%.4f' % r for r in res),
                                                                      float(sim.ws_cnt.sum(dtype=torch.int64).item()) / E, int(sim.status.max().item())), flush=True)


if __name__ == '__main__':
    if len(sys.argv) == 1:
        one()
    else:
        for name in sys.argv[1:]:
            lib = os.path.join(ROOT, 'gym_kilobots_amd', 'libkilobots_hip_%s.so' % name) if name != 'product' else os.path.join(ROOT, 'gym_kilobots_amd', 'libkilobots_hip.so')
            subprocess.call([sys.executable, os.path.abspath(__file__)], env=dict(os.environ, KB_HIP_LIB=lib))
