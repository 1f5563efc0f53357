#!/usr/bin/env python3
"""Phase shares of kb_step_kernel from the diagnostic build (-DKB_PROFILE): shader cycles of wave 0
per phase, averaged over envs.  Usage (GPU box):
    python -m gym_kilobots_amd.build  # product
    python -c "from gym_kilobots_amd import build; build.build_profile()"
    KB_HIP_LIB=gym_kilobots_amd/libkilobots_hip_prof.so python tools/phase_profile.py [--envs E --bots N]
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gym_kilobots_amd.sim import KilobotSim  # noqa: E402
from tests import scenes  # noqa: E402

PHASES = ['drive+grid', 'count pass', 'emit pass', 'islands+buckets', 'warm+velocity', 'store+integrate', 'position', 'reg load+depth (part of warm+velocity)']


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--envs', type=int, default=4096)
    ap.add_argument('--bots', type=int, default=1024)
    ap.add_argument('--warm', type=int, default=60)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--fused', type=int, default=1)
    ap.add_argument('--objects', type=int, default=0, help='cfg4: pushable objects per env')
    ap.add_argument('--boxes', action='store_true')
    ap.add_argument('--cluster', action='store_true', help='SimplePhototaxis swarm jamming around a light in the middle (one giant island): tools/cluster_probe.py')
    args = ap.parse_args()
    E, N = args.envs, args.bots
    assert 'prof' in os.environ.get('KB_HIP_LIB', ''), 'run with KB_HIP_LIB=.../libkilobots_hip_prof.so'
    okw = {}
    if args.objects and args.boxes:
        okw = dict(obj_shape=[1] * args.objects, obj_nverts=[4] * args.objects, obj_verts=[[[0.075 * 25.0, 0.075 * 25.0]]] * args.objects)
    if args.cluster:
        from gym_kilobots_amd import _native as nat
        sim = KilobotSim(E, N, nat.DRIVE_SIMPLE_PHOTOTAXIS, nat.LIGHT_CIRCULAR, light_radius=2.0, allow_sleep=0)
        sim.light_x.zero_()
        sim.light_y.zero_()
    else:
        sim = KilobotSim(E, N, num_objects=args.objects, allow_sleep=0, **okw)
    sim.status = torch.zeros(E + 40 * E, dtype=torch.int32, device=sim.device)   # status + stamp area
    sim._bind()
    xy1, th1 = scenes.lattice_spawn(8, N, seed=1000)
    reps = (E + 7) // 8
    sim.set_poses_m(np.tile(xy1, (reps, 1, 1))[:E], np.tile(th1, (reps, 1))[:E])
    if args.objects:
        sim.set_objects_m(np.tile(scenes.CFG4_OBJECTS[None, :args.objects], (E, 1, 1)))
    acts = []
    for k in range(8):
        a1 = scenes.random_actions(8, N, seed=2000 + k)
        if args.objects:
            a1[:, ::2, 0] = 0.01
            a1[:, ::2, 1] = 0.0
        acts.append(torch.from_numpy(np.tile(a1, (reps, 1, 1))[:E].copy()).cuda())
    for k in range(args.warm):
        sim.step(1, actions=None if args.cluster else acts[k % 8])
    torch.cuda.synchronize()
    sim.status[E:].zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for k in range(args.steps):
        sim.step(args.fused, actions=None if args.cluster else acts[k % 8])
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / args.steps
    raw = sim.status[E:].reshape(E, 40).double().cpu().numpy() / (args.steps * args.fused)
    st = raw[:, :8] * 16
    mean = st.mean(0)
    print('launch %.3f ms (%d substeps per launch); cycles per env-substep (wave 0), mean over %d envs:' % (ms, args.fused, E))
    for name, c in zip(PHASES, mean):
        print('  %-18s %9.0f  %5.1f%%' % (name, c, 100 * c / mean.sum()))
    print('  %-18s %9.0f' % ('total', mean.sum()))
    ex = raw[:, 8:13].mean(0)
    extra = raw[:, 13:16].mean(0) * 16
    print('  kernel start (per launch) %.0f, flatten+scan (part of islands+buckets) %.0f, write-back issue (thread 0) %.0f' % (extra[0], extra[1], raw[:, 22].mean() * 16))
    pre = raw[:, 16:19].mean(0) * 16
    print('  wave 0 before the barrier (own work): drive+grid %.0f, count pass %.0f, emit pass %.0f' % tuple(pre))
    sub_ = raw[:, 19:22].mean(0) * 16
    print('  wave 0 inside the register set-up, cumulative since the sort barrier: light load %.0f, + depth pass %.0f, + dealing %.0f' % tuple(sub_))
    fine = raw[:, 24:31].mean(0) * 16
    print('  since the register set-up: warm start %.0f, + 10 velocity sweeps of wave 0 %.0f (then the barrier)' % (fine[5], fine[6]))
    print('  since the integration: own position sweeps %.0f, + waiting for the slowest wave %.0f, + continuous-step candidates collected %.0f, '
          '+ own candidates processed %.0f, + every wave\'s %.0f (then copies, cell heads cleared, barrier)' % tuple(fine[:5]))
    b_ = raw[:, 31:36].mean(0) * 16
    print('  sorted bins, wave 0, cumulative since the drive barrier: boundary scan %.0f, + scatter and barrier %.0f, + slots settled %.0f; since the offset scan: contacts staged %.0f, + barrier %.0f' % tuple(b_))
    print('  wave 0 per substep: keys %.1f, depth rounds/sweep %.1f (deepest wave of the env %.1f), position sweeps %.2f, reg-path fraction %.2f' % (ex[0], ex[4], ex[1], ex[2], ex[3]))


if __name__ == '__main__':
    main()
