#!/usr/bin/env python3
"""bench.py -- kilobot-steps/s of the HIP world step on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

One "step" = ONE kb_step launch of ONE world substep (dt = 0.1 s; reference
kilobots_env.py:168-190) over the whole env batch with fresh per-bot actions, i.e. exactly the
unit SURVEY.md 8(d) prices at 48 algorithmic bytes per kilobot-step (state 20 B in + 20 B out +
command 8 B).  Workload at N = 1: BASELINE config 3, 4096 envs x 1024 kilobots (cfg3 of SURVEY 8d:
jittered 32 x 32 lattice, U([0,0.01] x [-pi/2,pi/2]) velocity-control actions), every env and every
action set an independent draw made on the device (torch generator keyed by seed and rank).

The timed state does not depend on --warmup: before the warm-up launches the scene is rolled forward
`--settle` substeps and then further until the contact population is stationary (contacts per env
within 1 % over 10 substeps); `contacts_per_env` in the JSON line is what the timed launches ran on.

Behind the timed region (N = 1, never mixed into `value`): `with_sleep_state` (the instantiation that carries Box2D's sleep
state), `fused_env_step` (10 substeps per launch) and `jammed_swarm` (the slow corner: the swarm jammed into one island).

N > 1: `--gpus N` without a torchrun environment starts N child processes itself (one per GPU,
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, fresh children -- the parent never touches a GPU);
under `python -m torch.distributed.run` the ranks are taken from the environment.  Every rank steps its
own 4096-env shard (weak scaling), there is no data-path collective; RCCL only gathers per-env episode
returns and the per-rank rates after the timed region.
"""
import argparse
import json
import os
import socket
import subprocess
import tempfile
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

ALGO_BYTES_PER_KILOBOT_STEP = 48.0
HBM_PEAK_GBS = 8000.0                  # MI355X_MICROARCH.md: HBM3E 8 TB/s
WORLD_SCALE = 25.0                     # body.py:7
ARENA_W, ARENA_H = 2.0, 1.5            # kilobots_env.py:19
CFG4_OBJECTS = np.array([[0.5, 0.35], [-0.5, 0.35], [-0.5, -0.35], [0.5, -0.35]])


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--settle', type=int, default=80, help='substeps of pre-roll before the warm-up count (then until stationary)')
    ap.add_argument('--seed', type=int, default=0)
    ap.add_argument('--envs', type=int, default=4096, help='envs per GPU')
    ap.add_argument('--bots', type=int, default=1024)
    ap.add_argument('--threads', type=int, default=0, help='workgroup size override')
    ap.add_argument('--objects', type=int, default=0, help='cfg4: 4 pushable discs per env (not the headline workload)')
    ap.add_argument('--boxes', action='store_true', help='with --objects: 0.15 m boxes (Quad) instead of discs')
    ap.add_argument('--sense', type=float, default=0.0, help='IR neighbour sensing radius in metres (0 = off; adds 4 B per kilobot-step)')
    ap.add_argument('--no-toi', action='store_true', help='disable the continuous step against the walls (A/B only)')
    ap.add_argument('--sleep', action='store_true', help='A/B only: the generic instantiation WITH the sleep state (b2World doSleep); the headline '
                    'workload commands every kilobot to move in every substep, where sleeping cannot change a bit (tests/test_sleeping.py), '
                    'so the fixed-size benchmark instantiation leaves the bookkeeping out')
    ap.add_argument('--vel-iters', type=int, default=10, help='solver velocity iterations (A/B only; the reference uses 10)')
    ap.add_argument('--pos-iters', type=int, default=10, help='solver position iterations (A/B only; the reference uses 10)')
    ap.add_argument('--arena', type=float, nargs=2, default=None, metavar=('W', 'H'), help='arena size in metres (A/B only; default 2.0 x 1.5)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-fused', action='store_true', help='skip the extra fused 10-substep launches after the timed region')
    ap.add_argument('--cpu-envs', type=int, default=64)
    ap.add_argument('--cpu-seconds', type=float, default=8.0)
    ap.add_argument('--traffic-json', default=os.path.join(ROOT, 'profiles', 'traffic_latest.json'))
    ap.add_argument('--selftest-launcher', action='store_true',
                    help='CPU-only check of the rank launcher and the gather (gloo); no simulation')
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------ launcher
def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(n, argv):
    """Start n fresh child processes of this script, one per GPU; the parent never initialises a GPU.
    Returns the worst child exit code.  Rank 0's stdout is this process's stdout (the ONE JSON line)."""
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        out = None if r == 0 else subprocess.DEVNULL
        # every rank's stderr goes to a temporary file: the tail of a failing rank's is forwarded with the exit message
        err = tempfile.TemporaryFile(mode='w+')
        p = subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env, stdout=out, stderr=err)
        p.kb_rank, p.kb_err = r, err
        procs.append(p)
    rc = 0
    deadline = time.time() + 3000
    pending = list(procs)
    while pending:
        for p in list(pending):
            c = p.poll()
            if c is None:
                continue
            pending.remove(p)
            p.kb_err.seek(0)
            tail = p.kb_err.read()[-4000:]
            if c != 0:
                if rc == 0:                # the first failure is the cause; the ranks terminated because of it only echo it
                    sys.stderr.write('bench.py: rank %d of %d exited with code %d; the end of its stderr:\n%s\n' % (p.kb_rank, n, c, tail))
                rc = rc or c
                for q in pending:          # one rank died: the others would wait in a collective for ever
                    q.terminate()
            elif p.kb_rank == 0 and tail:
                sys.stderr.write(tail)     # (warnings of a healthy run stay visible)
        if time.time() > deadline:
            for q in pending:
                q.kill()
            return rc or 124
        time.sleep(0.05)
    return rc


def selftest_launcher(args):
    """The launcher / rendezvous / gather path without a GPU (gloo): what tests/test_bench_launcher.py runs."""
    import torch
    import torch.distributed as dist
    from gym_kilobots_amd.dist import env_shard, gather_returns
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    if world > 1:
        dist.init_process_group('gloo')
    E = args.envs
    lo, hi = env_shard(E * world, rank, world)
    ret = torch.arange(lo, hi, dtype=torch.float32)
    allr = gather_returns(ret, dist if world > 1 else None)
    rate = torch.tensor([float(rank + 1)], dtype=torch.float64)
    rates = [torch.zeros_like(rate) for _ in range(world)]
    if world > 1:
        dist.all_gather(rates, rate)
        dist.barrier()
    else:
        rates = [rate]
    if rank == 0:
        ok = bool(torch.equal(allr, torch.arange(E * world, dtype=torch.float32)))
        print(json.dumps({'selftest': 'launcher', 'n_gpus': world, 'returns_gathered': int(allr.numel()),
                          'returns_in_global_env_order': ok, 'per_rank': [float(r.item()) for r in rates]}))
    if world > 1:
        dist.destroy_process_group()


# ------------------------------------------------------------------------------------------------ scene
def make_scene(torch, E, N, dev, seed, rank, objects, n_sets=8):
    """cfg3 scene of SURVEY 8(d) item 3, made on the device: every env its own jittered lattice (pitch 0.045 m, jitter
    +-0.004 m), headings U(-pi, pi); n_sets action sets U([0, 0.01] x [-pi/2, pi/2]) of shape [E, N, 2], all distinct."""
    g = torch.Generator(device=dev)
    g.manual_seed(1000003 * (seed + 1) + rank)
    side = int(np.ceil(np.sqrt(N)))
    idx = torch.arange(N, device=dev)
    pitch, jitter = 0.045, 0.004
    gx = ((idx % side).double() - (side - 1) / 2.0) * pitch
    gy = ((idx // side).double() - (side - 1) / 2.0) * pitch
    jit = (torch.rand(E, N, 2, generator=g, device=dev, dtype=torch.float64) * 2.0 - 1.0) * jitter
    x = ((gx[None] + jit[..., 0]) * WORLD_SCALE).float()
    y = ((gy[None] + jit[..., 1]) * WORLD_SCALE).float()
    th = ((torch.rand(E, N, generator=g, device=dev, dtype=torch.float64) * 2.0 - 1.0) * np.pi).float()
    acts = []
    for _ in range(n_sets):
        a = torch.rand(E, N, 2, generator=g, device=dev, dtype=torch.float32)
        a[..., 0] *= 0.01
        a[..., 1] = (a[..., 1] - 0.5) * float(np.pi)
        if objects:      # cfg4: every second bot drives straight ahead at full speed (sustained contact)
            a[:, ::2, 0] = 0.01
            a[:, ::2, 1] = 0.0
        acts.append(a.contiguous())
    return x, y, th, acts


def contacts_per_env(sim):
    """Mean number of touching contacts per env (every contact is owned by exactly one kilobot's warm-start entries)."""
    import torch
    return float(sim.ws_cnt.sum(dtype=torch.int64).item()) / sim.num_envs


def main():
    args = parse_args()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    if args.selftest_launcher:
        return selftest_launcher(args)

    import torch
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    dist = None
    # Rehearsal on a ONE-GPU box (the multi-rank path end to end, ranks sharing cuda:0): KB_BENCH_REHEARSAL=1 maps every
    # rank to device 0 and uses gloo for the collectives (RCCL refuses two ranks on one device).  Never set by the driver.
    rehearsal = os.environ.get('KB_BENCH_REHEARSAL') == '1'
    if rehearsal:
        local_rank = 0
    if world > 1:
        import torch.distributed as dist
        if rehearsal:
            dist.init_process_group('gloo')
        else:
            dist.init_process_group('nccl', device_id=torch.device('cuda:%d' % local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda:%d' % local_rank)
    cdev = torch.device('cpu') if rehearsal else dev          # where the few collective payloads live

    from gym_kilobots_amd.sim import KilobotSim
    from gym_kilobots_amd.dist import env_shard, gather_returns

    E, N = args.envs, args.bots
    total_envs = E * world
    lo, hi = env_shard(total_envs, rank, world)
    assert hi - lo == E
    okw = {}
    if args.objects and args.boxes:      # reference Quad(width=.15, height=.15): half extents in world units
        okw = dict(obj_shape=[1] * args.objects, obj_nverts=[4] * args.objects, obj_verts=[[[0.075 * 25.0, 0.075 * 25.0]]] * args.objects)
    if args.sense > 0.0:
        okw['sense_radius'] = args.sense
    if args.arena:
        okw.update(world_width=args.arena[0], world_height=args.arena[1])
    # The library default is the reference's world (doSleep=True, kilobots_env.py:45).  The headline times the instantiation without
    # the sleep state -- every kilobot is commanded in every substep, so sleeping cannot change a bit of this workload (tests/
    # test_sleeping.py) -- and reports the one that carries it beside it (`with_sleep_state`; --sleep runs everything on it).
    okw['allow_sleep'] = 1 if args.sleep else 0
    if args.vel_iters != 10 or args.pos_iters != 10:
        okw.update(vel_iters=args.vel_iters, pos_iters=args.pos_iters)
    sim = KilobotSim(E, N, device=dev, num_objects=args.objects, toi_walls=0 if args.no_toi else 1, **okw)
    if args.threads:
        sim.block_threads = args.threads
    x, y, th, actions = make_scene(torch, E, N, dev, args.seed, rank, args.objects)
    n_sets = len(actions)
    sim.x.copy_(x); sim.y.copy_(y); sim.theta.copy_(th)
    sim.forget_contacts()
    if args.objects:
        sim.set_objects_m(np.tile(CFG4_OBJECTS[None, :args.objects], (E, 1, 1)))
    x0 = sim.x.clone()
    y0 = sim.y.clone()

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- settle: the contact population builds up over ~60 substeps; time the stationary state, whatever --warmup is
    k_act = 0

    def run(n):
        nonlocal k_act
        for _ in range(n):
            sim.step(1, actions=actions[k_act % n_sets])
            k_act += 1
    run(args.settle)
    settled = args.settle
    c_prev = contacts_per_env(sim)
    for _ in range(20):
        run(10)
        settled += 10
        c_now = contacts_per_env(sim)
        stationary = abs(c_now - c_prev) <= 0.01 * max(c_now, 1.0)
        c_prev = c_now
        if stationary:
            break
    cpu_start = None
    if rank == 0 and not args.no_cpu_baseline and world == 1:
        nc = min(args.cpu_envs, E)
        cpu_start = dict(x=sim.x[:nc].cpu().numpy(), y=sim.y[:nc].cpu().numpy(), theta=sim.theta[:nc].cpu().numpy(),
                         actions=[a[:nc].cpu().numpy() for a in actions])
    run(args.warmup)
    c_before = contacts_per_env(sim)
    K = args.steps
    # The timed region: exactly K launches back to back between two HIP events on the stream the kernel is launched on
    # (KilobotSim launches on torch's current stream).  Event packets BETWEEN the launches cost the GPU 10 - 18 us per step
    # (0.337 against 0.319 ms), so the per-launch events are taken in a second, audit pass right behind the timed region.
    ev_a, ev_b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    barrier()
    t0 = time.perf_counter()
    ev_a.record()
    for k in range(K):
        sim.step(1, actions=actions[k_act % n_sets])
        k_act += 1
    ev_b.record()
    barrier()
    t1 = time.perf_counter()
    elapsed_local = t1 - t0
    elapsed = elapsed_local
    kern_ms = ev_a.elapsed_time(ev_b) / K          # average launch duration over the timed region
    c_after = contacts_per_env(sim)
    # audit pass: the same K launches again, each bracketed by its own events (first / last / min / max of the state that was timed)
    ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(K)]
    ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(K)]
    for k in range(K):
        ev0[k].record()
        sim.step(1, actions=actions[k_act % n_sets])
        ev1[k].record()
        k_act += 1
    torch.cuda.synchronize()
    launch_ms = [a.elapsed_time(b) for a, b in zip(ev0, ev1)]
    rates = [E * N * K / elapsed_local]
    if dist is not None:
        t = torch.tensor([elapsed], device=cdev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        r = torch.tensor([rates[0]], device=cdev, dtype=torch.float64)
        rl = [torch.zeros_like(r) for _ in range(world)]
        dist.all_gather(rl, r)
        rates = [float(v.item()) for v in rl]
    status = int(sim.status.max().item())

    # episode returns (negative mean displacement from spawn, a stand-in reward): gathered over RCCL
    ret = -torch.sqrt((sim.x - x0) ** 2 + (sim.y - y0) ** 2).mean(dim=1) / 25.0
    all_ret = gather_returns(ret.to(cdev), dist)

    # the same workload on the instantiation that carries Box2D's sleep state (b2World doSleep=True): reported next to the
    # headline, never mixed into it (the results of this workload are the same bits either way: tests/test_sleeping.py); it takes
    # over the state right behind the timed region, before the fused launches change the contact population
    with_sleep = None
    if rank == 0 and not args.no_fused and not args.sleep and not args.objects:
        okw2 = dict(okw, allow_sleep=1)
        sim2 = KilobotSim(E, N, device=dev, toi_walls=0 if args.no_toi else 1, **okw2)
        for name in ('x', 'y', 'theta', 'ws_key', 'ws_acc', 'ws_cnt'):
            getattr(sim2, name).copy_(getattr(sim, name))
        ns = max(10, min(40, K))
        for k in range(5):
            sim2.step(1, actions=actions[k % n_sets])
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for k in range(ns):
            sim2.step(1, actions=actions[k % n_sets])
        e1.record()
        torch.cuda.synchronize()
        sms = e0.elapsed_time(e1) / ns
        with_sleep = {'ms_per_launch': sms, 'kilobot_steps_per_s_one_gpu': E * N / (sms * 1e-3), 'launches': ns,
                      'bytes_per_kilobot_step': 56, 'asleep_at_the_end': int((sim2.sleep_time < 0).sum().item()),
                      'status_flags': int(sim2.status.max().item())}
        sim2.close()
        del sim2
    # fused variant: one launch = one env.step = 10 substeps, state LDS-resident in between
    fused = None
    if rank == 0 and not args.no_fused:
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

    # the slow corner, reported next to the headline and never mixed into it: the same swarm driven at a light in the middle
    # of the arena (SimplePhototaxis) jams into ONE island of ~ 2 800 contacts, which the whole workgroup sweeps level by level
    # on depth-sorted records in the global staging slice (tools/cluster_probe.py, tests/test_parity_gpu.py::test_jammed_swarm_*)
    jammed = None
    plain = not (args.objects or args.sleep or args.sense > 0.0 or args.threads or args.no_toi or args.arena
                 or args.vel_iters != 10 or args.pos_iters != 10)
    if rank == 0 and not args.no_fused and plain and world == 1:
        from gym_kilobots_amd import _native as nat
        # (as many envs as the headline: with the 1024 of rounds 2 - 3m a third of them ran alone on their CU behind the first 768,
        #  1.8e9 instead of 2.2e9 on the same kernel)
        Ej, settle_j, nj = E, 350, 20
        sim3 = KilobotSim(Ej, N, nat.DRIVE_SIMPLE_PHOTOTAXIS, nat.LIGHT_CIRCULAR, device=dev, light_radius=2.0, allow_sleep=0)
        sim3.x.copy_(x0[:Ej]); sim3.y.copy_(y0[:Ej]); sim3.theta.copy_(th[:Ej])
        sim3.forget_contacts()
        sim3.light_x.zero_(); sim3.light_y.zero_()
        for _ in range(settle_j):
            sim3.step(1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(nj):
            sim3.step(1)
        e1.record()
        torch.cuda.synchronize()
        jms = e0.elapsed_time(e1) / nj
        jammed = {'workload': '%d envs x %d SimplePhototaxis kilobots around a CircularGradientLight in the middle of the arena, '
                              '%d substeps after the lattice spawn' % (Ej, N, settle_j),
                  'ms_per_launch': jms, 'kilobot_steps_per_s_one_gpu': Ej * N / (jms * 1e-3), 'launches': nj,
                  'contacts_per_env': contacts_per_env(sim3), 'status_flags': int(sim3.status.max().item())}
        sim3.close()
        del sim3

    # the other single-GPU configurations of BASELINE.json, each with its own launch time and roofline fraction -- reported
    # next to the headline, never mixed into `value`: cfg2 (256 envs x 64 kilobots) and cfg4 (cfg3 + 4 pushable discs, every
    # second kilobot ramming ahead) right behind the settle AND 500 substeps later (the rammers pile up against the walls)
    cfg2 = cfg4 = None
    if rank == 0 and not args.no_fused and plain and world == 1:
        def timed(sim_, acts_, n_):
            e0_, e1_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0_.record()
            for k_ in range(n_):
                sim_.step(1, actions=acts_[k_ % len(acts_)])
            e1_.record()
            torch.cuda.synchronize()
            return e0_.elapsed_time(e1_) / n_

        def leg(sim_, ms_, what):
            Ee, Nn = sim_.num_envs, sim_.num_bots
            return {'workload': what, 'ms_per_launch': ms_, 'kilobot_steps_per_s_one_gpu': Ee * Nn / (ms_ * 1e-3),
                    'contacts_per_env': contacts_per_env(sim_), 'status_flags': int(sim_.status.max().item()),
                    'roofline_frac': ALGO_BYTES_PER_KILOBOT_STEP * Ee * Nn / (ms_ * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    'resident_envs_per_cu': sim_.resident_envs_per_cu, 'workgroup_threads': sim_.block_threads}
        # cfg2: Gaussian spawn sigma 0.15 m, 100 substeps of warm-up, 400 timed launches
        E2, N2 = 256, 64
        s2 = KilobotSim(E2, N2, device=dev, allow_sleep=0)
        g2 = torch.Generator(device=dev); g2.manual_seed(4242 + args.seed)
        s2.x.copy_((torch.randn(E2, N2, generator=g2, device=dev) * 0.15 * WORLD_SCALE).clamp_(-0.98 * WORLD_SCALE, 0.98 * WORLD_SCALE))
        s2.y.copy_((torch.randn(E2, N2, generator=g2, device=dev) * 0.15 * WORLD_SCALE).clamp_(-0.73 * WORLD_SCALE, 0.73 * WORLD_SCALE))
        s2.theta.copy_((torch.rand(E2, N2, generator=g2, device=dev) * 2.0 - 1.0) * float(np.pi))
        s2.forget_contacts()
        a2 = []
        for _ in range(4):
            a_ = torch.rand(E2, N2, 2, generator=g2, device=dev)
            a_[..., 0] *= 0.01
            a_[..., 1] = (a_[..., 1] - 0.5) * float(np.pi)
            a2.append(a_.contiguous())
        timed(s2, a2, 100)
        cfg2 = leg(s2, timed(s2, a2, 400), 'cfg2: 256 envs x 64 kilobots, Gaussian spawn (sigma 0.15 m), random velocity actions, 1 substep per launch, 400 launches after 100 of warm-up')
        s2.close()
        del s2
        # cfg4: the headline scene + 4 discs of radius 0.075 m at (+-0.5, +-0.35) m, every second kilobot at full speed straight ahead
        s4 = KilobotSim(E, N, device=dev, num_objects=4, allow_sleep=0)
        x4, y4, th4, a4 = make_scene(torch, E, N, dev, args.seed, rank, 4)
        s4.x.copy_(x4); s4.y.copy_(y4); s4.theta.copy_(th4)
        s4.forget_contacts()
        s4.set_objects_m(np.tile(CFG4_OBJECTS[None, :4], (E, 1, 1)))
        timed(s4, a4, settled)
        ms_a = timed(s4, a4, 40)
        c_a = leg(s4, ms_a, 'cfg4: cfg3 + 4 pushable discs, every second kilobot ramming ahead; 40 launches right behind %d substeps of settling' % settled)
        timed(s4, a4, max(0, 500 - settled - 40))
        ms_b = timed(s4, a4, 40)
        cfg4 = {'after_settle': c_a, 'after_500_substeps': leg(s4, ms_b, 'the same scene, 40 launches after 500 substeps')}
        s4.close()
        del s4

    if rank != 0:
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    algo_bytes = ALGO_BYTES_PER_KILOBOT_STEP + (4.0 if args.sense > 0.0 else 0.0)   # + neighbour count u32
    value = total_envs * N * K / elapsed
    achieved = algo_bytes * E * N / (kern_ms * 1e-3) / 1e9
    traffic, traffic_source = None, None
    if os.path.exists(args.traffic_json) and not args.objects and args.sense <= 0.0:
        try:
            tj = json.load(open(args.traffic_json))
            if tj.get('envs') == E and tj.get('bots') == N:
                traffic = tj.get('hbm_bytes_per_launch')
                traffic_source = ('%s: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/measure_round.sh on this workload '
                                  '(corrected as MI355X_MICROARCH.md prescribes); NOT measured in this run'
                                  % os.path.relpath(args.traffic_json, ROOT))
        except Exception:
            traffic = None
    out = {
        'metric': 'kilobot-steps/sec', 'value': value, 'unit': 'kilobot-steps/s', 'n_gpus': world,
        'steps': K, 'warmup': args.warmup, 'ms_per_step': elapsed / K * 1e3, 'higher_is_better': True,
        'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': ('cfg4 (+%d pushable %s): ' % (args.objects, 'boxes' if args.boxes else 'discs') if args.objects else '') + 'cfg3: %d envs x %d kilobots per GPU, empty arena 2.0x1.5 m, jittered lattice spawn, '
                               'random velocity-control actions every substep, every env and action set an independent draw; '
                               '1 step = 1 world substep (dt 0.1 s) in 1 launch' % (E, N),
                   'envs_per_gpu': E, 'bots': N, 'total_envs': total_envs, 'substeps_per_launch': 1,
                   'parallelism': 'env-shard x%d' % world + (' (REHEARSAL: all ranks on one GPU, gloo)' if rehearsal else ''), 'workgroup_threads': sim.block_threads,
                   'lds_bytes_per_env': sim.lds_bytes, 'resident_envs_per_cu': sim.resident_envs_per_cu, 'seed': args.seed,
                   'settle_substeps': settled, 'settle_stationary': bool(stationary),
                   'sense_radius_m': args.sense,
                   'sleep_state': bool(args.sleep),
                   'sleep_note': 'b2World(doSleep=True) bookkeeping ' + ('carried (generic instantiation)' if args.sleep else
                                 'left out: every kilobot is commanded to move in every substep, so sleeping cannot change a bit '
                                 '(proved + tested: tests/test_sleeping.py); `--sleep` times the instantiation that carries it')},
        'contacts_per_env': c_before, 'contacts_per_env_after': c_after,
        'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                     'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic, 'traffic_source': traffic_source,
                     'kernel': 'kb_step_kernel', 'avg_launch_ms': kern_ms,
                     'avg_launch_source': 'one HIP event pair around the K back-to-back launches of the timed region / K',
                     'audit_pass': 'the K launches right behind the timed region, one event pair each: avg %.4f ms' % float(np.mean(launch_ms)),
                     'first_launch_ms': launch_ms[0], 'last_launch_ms': launch_ms[-1],
                     'min_launch_ms': float(np.min(launch_ms)), 'max_launch_ms': float(np.max(launch_ms)),
                     'algorithmic_bytes_per_launch': algo_bytes * E * N},
        'per_rank_kilobot_steps_per_s': rates,
        'fused_env_step': fused,
        'with_sleep_state': with_sleep,
        'jammed_swarm': jammed,
        'cfg2': cfg2,
        'cfg4': cfg4,
        'status_flags': status,
        'returns_gathered': int(all_ret.numel()),
        'returns_expected': int(total_envs),
        'collective_world_size': int(dist.get_world_size()) if dist is not None else 1,
        'collective_backend': (dist.get_backend() if dist is not None else None),
    }
    if cpu_start is not None:
        out['cpu_baseline'] = cpu_baseline(args, N, cpu_start)
    print(json.dumps(out))
    sys.stdout.flush()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


# ------------------------------------------------------------------------------------------------ CPU legs
def _host_cores():
    # the GPU box gives one job a 16-CPU share whatever os.cpu_count() says
    try:
        aff = len(os.sched_getaffinity(0))
    except AttributeError:
        aff = os.cpu_count() or 1
    return max(1, min(aff, 16)), aff


def _time_oracle(o, acts, threads, seconds, warm=2):
    o.set_actions(acts[0])
    o.step(warm, threads=threads)
    t0 = time.perf_counter()
    n = 0
    while True:
        o.set_actions(acts[n % len(acts)])
        o.step(1, threads=threads)
        n += 1
        if time.perf_counter() - t0 > seconds:
            break
    return n, time.perf_counter() - t0


def cpu_baseline(args, N, start):
    """The oracle (C restatement of the reference path, oracle/kb_oracle.c) on the host cores: SURVEY 8(d)'s CPU legs.
    A reported baseline, never the target."""
    from oracle import oracle as O
    cores, aff = _host_cores()
    Ec = start['x'].shape[0]

    def slice_sim(ne):
        o = O.OracleSim(O.default_config(ne, N))
        o.x[...] = start['x'][:ne]; o.y[...] = start['y'][:ne]; o.theta[...] = start['theta'][:ne]
        return o, [a[:ne] for a in start['actions']]
    # (i) all host cores, the bench workload's first Ec envs in their settled state
    o, acts = slice_sim(Ec)
    n, dt = _time_oracle(o, acts, cores, args.cpu_seconds)
    res = {'value': Ec * N * n / dt, 'unit': 'kilobot-steps/s', 'cores': cores, 'kind': 'port',
           'sample': '%d envs x %d bots x %d substeps of the same workload in its settled state (oracle/kb_oracle.c, OpenMP over envs), '
                     '%.1f s; scales linearly in envs (independent worlds)' % (Ec, N, n, dt),
           'nproc': os.cpu_count(), 'affinity_cpus': aff}
    # (ii) single thread, 4 envs of the same workload
    ns = min(4, Ec)
    o1, acts1 = slice_sim(ns)
    n1, dt1 = _time_oracle(o1, acts1, 1, max(2.0, args.cpu_seconds / 2))
    res['single_thread'] = {'value': ns * N * n1 / dt1, 'cores': 1,
                            'sample': '%d envs x %d bots x %d substeps, %.1f s' % (ns, N, n1, dt1)}
    # (iii) cfg1 in full: 1 env x 16 kilobots, Gaussian spawn sigma 0.1, theta 0, 100 env.steps of 10 substeps
    rng = np.random.RandomState(args.seed)

    def gaussian(E_, N_, sigma, random_theta):
        xy = rng.normal(scale=sigma, size=(E_, N_, 2))
        lo = np.array([-ARENA_W / 2, -ARENA_H / 2]) + 0.02
        hi = np.array([ARENA_W / 2, ARENA_H / 2]) - 0.02
        xy = np.minimum(np.maximum(xy, lo), hi)
        th = rng.uniform(-np.pi, np.pi, size=(E_, N_)) if random_theta else np.zeros((E_, N_))
        return xy, th

    def rand_actions(E_, N_):
        return rng.uniform([0.0, -np.pi / 2], [0.01, np.pi / 2], size=(E_, N_, 2)).astype(np.float32)
    o = O.OracleSim(O.default_config(1, 16))
    xy, th = gaussian(1, 16, 0.1, False)
    o.set_poses_m(xy, th)
    o.step(1, flags=O.STEP_NO_DRIVE)
    a1 = [rand_actions(1, 16) for _ in range(100)]
    t0 = time.perf_counter()
    for a in a1:
        o.set_actions(a)
        o.step(10)
    dt = time.perf_counter() - t0
    res['cfg1'] = {'value': 16 * 1000 / dt, 'cores': 1, 'sample': '1 env x 16 bots, 100 env.steps (1000 substeps) in full, %.4f s' % dt}
    # (iv) cfg2 in full: 256 x 64, sigma 0.15, random headings, 100 warm-up + 1000 timed substeps, 1 thread and all cores
    res['cfg2'] = {}
    for tag, thr in (('single_thread', 1), ('all_cores', cores)):
        rng = np.random.RandomState(args.seed + 1)
        o = O.OracleSim(O.default_config(256, 64))
        xy, th = gaussian(256, 64, 0.15, True)
        o.set_poses_m(xy, th)
        a2 = [rand_actions(256, 64) for _ in range(8)]
        for k in range(100):
            o.set_actions(a2[k % 8])
            o.step(1, threads=thr)
        t0 = time.perf_counter()
        for k in range(1000):
            o.set_actions(a2[k % 8])
            o.step(1, threads=thr)
        dt = time.perf_counter() - t0
        res['cfg2'][tag] = {'value': 256 * 64 * 1000 / dt, 'cores': thr,
                            'sample': '256 envs x 64 bots, 1000 substeps after 100 warm-up in full, %.2f s' % dt}
    res['reference_python_glue_floor'] = {
        'value': 1.25e5, 'range': [1.0e5, 1.5e5], 'unit': 'kilobot-steps/s per core',
        'note': 'BASELINE.md section 2: cost of the reference Python drive law + get_state alone (Box2D solver time excluded), '
                'measured in the survey container on one 2.1 GHz Xeon core; NOT measured in this run (the reference cannot travel to the GPU box)'}
    return res


if __name__ == '__main__':
    main()
