#!/usr/bin/env python3
"""Instruction and time budget of the benchmark kernel, phase by phase.

Measurement builds of the benchmark instantiation (-DKB_ABLATE=k -DKB_ONLY_BENCH) end the kernel behind phase k without
writing anything back; run on the SAME settled cfg3 state under rocprofv3, the differences between consecutive builds
are what each phase costs in wave instructions (SQ_INSTS_VALU / _SALU / _LDS) and in launch time with the CU loaded
exactly as in the product (three resident envs, all phases of other envs in flight).

    python tools/phase_ablation.py --build            # here (no GPU): the 15 variant libraries, in parallel
    python tools/phase_ablation.py --settle           # GPU box, product library: settle the scene, save the state
    rocprofv3 --pmc ... -- python3 tools/phase_ablation.py --run k    # GPU box, variant k (KB_HIP_LIB is set by --all)
    python tools/phase_ablation.py --all              # GPU box: settle, then every variant under rocprofv3; prints the table
"""
import argparse
import csv
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PHASES = ['(launch + state load)', 'drive law + bin counters', 'bins sorted (scan, scatter, settle)', 'find owned contacts', 'packed-list offsets (scan)',
          'stage contacts (owner)', 'label: impulses + island hooking', 'flatten roots', 'census + placement + group by wave',
          'register set-up (load, depth, dealing)', 'warm start', '10 velocity sweeps', 'store + integrate', 'position sweeps',
          'continuous step', 'end of substep + write-back']
STATE = '/tmp/kb_abl_state.pt'


def lib(k):
    return os.path.join(ROOT, 'gym_kilobots_amd', 'libkilobots_hip_abl%d.so' % k)


def build_one(k):
    from gym_kilobots_amd import build as B
    B.build_variant('abl%d' % k, ['-DKB_ABLATE=%d' % k, '-DKB_ONLY_BENCH'], only=['d0'], standalone=True)
    return k


def settle():
    import torch
    import bench
    from gym_kilobots_amd.sim import KilobotSim
    dev = torch.device('cuda:0')
    E, N = 4096, 1024
    sim = KilobotSim(E, N, device=dev, allow_sleep=0)
    x, y, th, acts = bench.make_scene(torch, E, N, dev, 0, 0, 0)
    sim.x.copy_(x); sim.y.copy_(y); sim.theta.copy_(th)
    sim.forget_contacts()
    for k in range(125):
        sim.step(1, actions=acts[k % 8])
    torch.cuda.synchronize()
    torch.save({f: getattr(sim, f).cpu() for f in ('x', 'y', 'theta', 'v', 'w', 'ws_key', 'ws_acc', 'ws_cnt')} | {'actions': acts[5].cpu()}, STATE)
    print('settled: %.1f contacts per env' % (float(sim.ws_cnt.sum(dtype=torch.int64).item()) / E))


def run(k, launches):
    import torch
    from gym_kilobots_amd.sim import KilobotSim
    dev = torch.device('cuda:0')
    st = torch.load(STATE)
    sim = KilobotSim(4096, 1024, device=dev, allow_sleep=0)
    for f in ('x', 'y', 'theta', 'v', 'w', 'ws_key', 'ws_acc', 'ws_cnt'):
        getattr(sim, f).copy_(st[f].to(dev))
    a = st['actions'].to(dev)
    for _ in range(launches):
        sim.step(1, actions=a)
    torch.cuda.synchronize()


def collect(outdir):
    vals, durs = {}, []
    for f in glob.glob(os.path.join(outdir, '*', '*counter_collection.csv')):
        per = {}
        for r in csv.DictReader(open(f)):
            if 'kb_step_kernel' in r['Kernel_Name']:
                per.setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
        for c, v in per.items():
            vals[c] = sum(v[2:]) / max(1, len(v[2:]))
    for f in glob.glob(os.path.join(outdir, '*', '*kernel_trace.csv')):
        for r in csv.DictReader(open(f)):
            if 'kb_step_kernel' in r['Kernel_Name']:
                durs.append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-3)
    return vals, (sum(durs[2:]) / max(1, len(durs[2:])) if durs else float('nan'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--build', action='store_true')
    ap.add_argument('--settle', action='store_true')
    ap.add_argument('--run', type=int, default=-1)
    ap.add_argument('--all', action='store_true')
    ap.add_argument('--launches', type=int, default=8)
    ap.add_argument('--out', default=os.path.join(ROOT, 'gpurun_out', 'ablation'))
    args = ap.parse_args()
    ks = list(range(1, 15)) + [99]
    if args.build:
        from concurrent.futures import ProcessPoolExecutor
        with ProcessPoolExecutor(max_workers=4) as ex:
            for k in ex.map(build_one, ks):
                print('built variant', k, flush=True)
        return
    if args.settle:
        return settle()
    if args.run >= 0:
        return run(args.run, args.launches)
    if args.all:
        os.makedirs(args.out, exist_ok=True)
        subprocess.check_call([sys.executable, os.path.abspath(__file__), '--settle'])
        rows = []
        counters = ['SQ_INSTS_VALU', 'SQ_INSTS_SALU', 'SQ_INSTS_LDS', 'SQ_WAVE_CYCLES', 'SQ_WAIT_ANY', 'SQ_INSTS_FLAT']
        for k in ks:
            od = os.path.join(args.out, 'k%d' % k)
            subprocess.call(['rm', '-rf', od])
            env = dict(os.environ, KB_HIP_LIB=lib(k), TMPDIR='/tmp')
            cmd = ['rocprofv3', '--pmc'] + counters + ['--kernel-trace', '--output-format', 'csv', '-d', od, '--',
                                                      'python3', os.path.abspath(__file__), '--run', str(k), '--launches', str(args.launches)]
            with open(os.path.join(args.out, 'k%d.log' % k), 'w') as lg:
                subprocess.check_call(cmd, env=env, cwd='/tmp', stdout=lg, stderr=subprocess.STDOUT)
            vals, dur = collect(od)
            rows.append((k, vals, dur))
            subprocess.call(['rm', '-rf', od])
            print('variant %d: %.1f us' % (k, dur), flush=True)
        print('%-44s %9s %9s %9s %9s %9s' % ('phase (cost = difference to the build before)', 'us', 'VALU M', 'SALU M', 'LDS M', 'FLAT M'))
        prev = ({c: 0.0 for c in counters}, 0.0)
        for (k, vals, dur), name in zip(rows, PHASES[1:]):
            d = {c: vals.get(c, 0.0) - prev[0].get(c, 0.0) for c in counters}
            print('%-44s %9.1f %9.2f %9.2f %9.2f %9.2f' % (name, dur - prev[1], d['SQ_INSTS_VALU'] / 1e6, d['SQ_INSTS_SALU'] / 1e6, d['SQ_INSTS_LDS'] / 1e6, d['SQ_INSTS_FLAT'] / 1e6))
            prev = (vals, dur)
        print('%-44s %9.1f %9.2f %9.2f %9.2f %9.2f' % ('total', prev[1], prev[0].get('SQ_INSTS_VALU', 0) / 1e6, prev[0].get('SQ_INSTS_SALU', 0) / 1e6,
                                                       prev[0].get('SQ_INSTS_LDS', 0) / 1e6, prev[0].get('SQ_INSTS_FLAT', 0) / 1e6))


if __name__ == '__main__':
    main()
