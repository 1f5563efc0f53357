"""bench.py --gpus N starts N ranks itself: the launcher, the rendezvous and the gather of
per-env returns are exercised on CPU with gloo (world 2 and 3); the simulation itself needs a GPU and is not run."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(n, envs):
    env = dict(os.environ)
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', str(n), '--envs', str(envs),
                        '--selftest-launcher'], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=240)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    # (gloo itself prints a '[Gloo] Rank 0 is connected ...' line on stdout; RCCL does not)
    lines = [l for l in p.stdout.decode().splitlines() if l.strip() and not l.startswith('[Gloo]')]
    assert len(lines) == 1, lines          # rank 0 prints ONE line, the other ranks none
    return json.loads(lines[0])


def test_launcher_spawns_one_rank_per_gpu():
    d = _run(2, 5)
    assert d['n_gpus'] == 2 and d['returns_gathered'] == 10 and d['returns_in_global_env_order']
    assert d['per_rank'] == [1.0, 2.0]


def test_launcher_three_ranks():
    d = _run(3, 4)
    assert d['n_gpus'] == 3 and d['returns_gathered'] == 12 and d['returns_in_global_env_order']


def test_single_rank_needs_no_launcher():
    d = _run(1, 7)
    assert d['n_gpus'] == 1 and d['returns_gathered'] == 7


def test_failing_rank_fails_the_launch():
    env = dict(os.environ)
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    # without --selftest-launcher the children need a GPU: on a CPU box they fail, and the launcher must report it
    import torch
    if torch.cuda.is_available():
        return
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--envs', '4', '--bots', '16',
                        '--steps', '1', '--warmup', '0', '--no-cpu-baseline'], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=240)
    assert p.returncode != 0
    # ... and say which rank died and why (the tail of its stderr travels with the exit message)
    err = p.stderr.decode()
    assert 'exited with code' in err and 'the end of its stderr' in err and ('Error' in err or 'error' in err), err[-1500:]
