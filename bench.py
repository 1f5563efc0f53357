#!/usr/bin/env python3
"""bench.py -- kilobot-steps/s of the HIP world step on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

One "step" = ONE kb_step launch of ONE world substep (dt = 0.1 s; reference
kilobots_env.py:168-190) over the whole env batch with fresh per-bot actions, i.e. exactly the
unit SURVEY.md 8(d) prices at 48 algorithmic bytes per kilobot-step (state 20 B in + 20 B out +
command 8 B).  Workload at N = 1: BASELINE config 3, 4096 envs x 1024 kilobots (cfg3 of SURVEY 8d:
jittered 32 x 32 lattice, U([0,0.01] x [-pi/2,pi/2]) velocity-control actions).  N > 1: one process
per GPU, the same 4096-env shard on every rank (weak scaling), no data-path collective; RCCL only
gathers per-env episode returns after the timed region.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

ALGO_BYTES_PER_KILOBOT_STEP = 48.0
HBM_PEAK_GBS = 8000.0                  # MI355X_MICROARCH.md: HBM3E 8 TB/s


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=60)
    ap.add_argument('--envs', type=int, default=4096, help='envs per GPU')
    ap.add_argument('--bots', type=int, default=1024)
    ap.add_argument('--threads', type=int, default=0, help='workgroup size override')
    ap.add_argument('--objects', type=int, default=0, help='cfg4: 4 pushable discs per env (not the headline workload)')
    ap.add_argument('--boxes', action='store_true', help='with --objects: 0.15 m boxes (Quad) instead of discs')
    ap.add_argument('--no-toi', action='store_true', help='disable the continuous step against the walls (A/B only)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-envs', type=int, default=64)
    ap.add_argument('--traffic-json', default=os.path.join(ROOT, 'profiles', 'traffic_latest.json'))
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group('nccl', device_id=torch.device('cuda:%d' % local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda:%d' % local_rank)

    from gym_kilobots_amd.sim import KilobotSim
    from gym_kilobots_amd.dist import env_shard, gather_returns
    from tests import scenes

    E, N = args.envs, args.bots
    total_envs = E * world
    lo, hi = env_shard(total_envs, rank, world)
    assert hi - lo == E
    # synthetic scene: every env is an independent draw; seeds are keyed by global env index blocks
    base = 8
    xy1, th1 = scenes.lattice_spawn(base, N, seed=1000 + rank)
    reps = (E + base - 1) // base
    xy = np.tile(xy1, (reps, 1, 1))[:E]
    th = np.tile(th1, (reps, 1))[:E]
    okw = {}
    if args.objects and args.boxes:      # reference Quad(width=.15, height=.15): half extents in world units
        okw = dict(obj_shape=[1] * args.objects, obj_nverts=[4] * args.objects, obj_verts=[[[0.075 * 25.0, 0.075 * 25.0]]] * args.objects)
    sim = KilobotSim(E, N, device=dev, num_objects=args.objects, toi_walls=0 if args.no_toi else 1, **okw)
    if args.threads:
        sim.block_threads = args.threads
    sim.set_poses_m(xy, th)
    if args.objects:
        sim.set_objects_m(np.tile(scenes.CFG4_OBJECTS[None, :args.objects], (E, 1, 1)))
    n_sets = 8
    actions = []
    for k in range(n_sets):
        a1 = scenes.random_actions(base, N, seed=2000 + 10 * rank + k)
        if args.objects:      # cfg4: every second bot drives straight ahead at full speed (sustained contact)
            a1[:, ::2, 0] = 0.01
            a1[:, ::2, 1] = 0.0
        actions.append(torch.from_numpy(np.tile(a1, (reps, 1, 1))[:E].copy()).to(dev))
    x0 = sim.x.clone()
    y0 = sim.y.clone()

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for k in range(args.warmup):
        sim.step(1, actions=actions[k % n_sets])
    K = args.steps
    ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(K)]
    ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(K)]
    barrier()
    t0 = time.perf_counter()
    for k in range(K):
        ev0[k].record()
        sim.step(1, actions=actions[k % n_sets])
        ev1[k].record()
    barrier()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in zip(ev0, ev1)]))
    if dist is not None:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    status = int(sim.status.max().item())

    # episode returns (negative mean displacement from spawn, a stand-in reward): gathered over RCCL
    ret = -torch.sqrt((sim.x - x0) ** 2 + (sim.y - y0) ** 2).mean(dim=1) / 25.0
    all_ret = gather_returns(ret, dist)

    # fused variant: one launch = one env.step = 10 substeps, state LDS-resident in between
    fused = None
    if rank == 0:
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        nf = max(2, min(10, K // 10))
        sim.step(10, actions=actions[0])
        e0.record()
        for k in range(nf):
            sim.step(10, actions=actions[k % n_sets])
        e1.record()
        torch.cuda.synchronize()
        fms = e0.elapsed_time(e1) / nf
        fused = {'substeps_per_launch': 10, 'ms_per_launch': fms,
                 'kilobot_steps_per_s_one_gpu': E * N * 10 / (fms * 1e-3)}

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    value = total_envs * N * K / elapsed
    achieved = ALGO_BYTES_PER_KILOBOT_STEP * E * N / (kern_ms * 1e-3) / 1e9
    traffic = None
    if os.path.exists(args.traffic_json):
        try:
            tj = json.load(open(args.traffic_json))
            if tj.get('envs') == E and tj.get('bots') == N:
                traffic = tj.get('hbm_bytes_per_launch')
        except Exception:
            traffic = None
    out = {
        'metric': 'kilobot-steps/sec', 'value': value, 'unit': 'kilobot-steps/s', 'n_gpus': world,
        'steps': K, 'warmup': args.warmup, 'ms_per_step': elapsed / K * 1e3, 'higher_is_better': True,
        'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': ('cfg4 (+%d pushable %s): ' % (args.objects, 'boxes' if args.boxes else 'discs') if args.objects else '') + 'cfg3: %d envs x %d kilobots per GPU, empty arena 2.0x1.5 m, jittered lattice spawn, '
                               'random velocity-control actions every substep; 1 step = 1 world substep (dt 0.1 s) in 1 launch'
                               % (E, N),
                   'envs_per_gpu': E, 'bots': N, 'total_envs': total_envs, 'substeps_per_launch': 1,
                   'parallelism': 'env-shard x%d' % world, 'workgroup_threads': sim.block_threads,
                   'lds_bytes_per_env': sim.lds_bytes},
        'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                     'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic,
                     'kernel': 'kb_step_kernel', 'avg_launch_ms': kern_ms,
                     'algorithmic_bytes_per_launch': ALGO_BYTES_PER_KILOBOT_STEP * E * N},
        'fused_env_step': fused,
        'status_flags': status,
        'returns_gathered': int(all_ret.numel()),
    }
    if not args.no_cpu_baseline and world == 1:
        out['cpu_baseline'] = cpu_baseline(args, N, xy1, th1)
    print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


def cpu_baseline(args, N, xy1, th1):
    """The oracle (C restatement of the reference path) on the host cores, same workload, bounded sample."""
    from oracle import oracle as O
    from tests import scenes
    # the GPU box gives one job a 16-CPU share whatever os.cpu_count() says
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 16))
    Ec = args.cpu_envs
    base = xy1.shape[0]
    reps = (Ec + base - 1) // base
    o = O.OracleSim(O.default_config(Ec, N))
    o.set_poses_m(np.tile(xy1, (reps, 1, 1))[:Ec], np.tile(th1, (reps, 1))[:Ec])
    acts = [np.tile(scenes.random_actions(base, N, seed=2000 + k), (reps, 1, 1))[:Ec] for k in range(8)]
    o.set_actions(acts[0])
    o.step(2, threads=cores)
    t0 = time.perf_counter()
    n = 0
    while True:
        o.set_actions(acts[n % 8])
        o.step(1, threads=cores)
        n += 1
        if time.perf_counter() - t0 > 12.0:
            break
    dt = time.perf_counter() - t0
    return {'value': Ec * N * n / dt, 'unit': 'kilobot-steps/s', 'cores': cores, 'kind': 'port',
            'sample': '%d envs x %d bots x %d substeps of the same workload (oracle/kb_oracle.c, OpenMP over envs), %.1f s'
                      % (Ec, N, n, dt)}


if __name__ == '__main__':
    main()
