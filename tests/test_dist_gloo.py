"""The multi-GPU path on CPU: world_size-2 gloo processes, env-axis sharding with no data-path
collective, one all-gather of per-env episode returns."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gym_kilobots_amd import dist as kdist
from oracle import oracle as O
from tests import scenes

TOTAL_ENVS, N, STEPS = 5, 16, 3      # odd on purpose: shards of unequal size


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _episode(xy, th, acts):
    from tests.oracle_backend import OracleBackend
    from gym_kilobots_amd.envs import BatchedKilobotsEnv
    E = xy.shape[0]
    env = BatchedKilobotsEnv(E, N, sim_factory=OracleBackend,
                             reward_fn=lambda prev, a, obs: (obs[..., :2] - prev[..., :2]).norm(dim=-1).sum(-1))
    env.reset(poses=(xy, th))
    for a in acts:
        env.step(torch.from_numpy(a))
    return env


def _worker(rank, world, port, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    xy, th = scenes.gaussian_spawn(TOTAL_ENVS, N, sigma=0.08, seed=1)
    acts = [scenes.random_actions(TOTAL_ENVS, N, seed=10 + k) for k in range(STEPS)]
    lo, hi = kdist.env_shard(TOTAL_ENVS, rank, world)
    env = _episode(xy[lo:hi], th[lo:hi], [a[lo:hi] for a in acts])
    allr = env.gather_episode_returns(dist)
    torch.save({'returns': allr, 'poses': env.sim.poses(), 'range': (lo, hi)}, os.path.join(out, 'r%d.pt' % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_equals_single_process(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    xy, th = scenes.gaussian_spawn(TOTAL_ENVS, N, sigma=0.08, seed=1)
    acts = [scenes.random_actions(TOTAL_ENVS, N, seed=10 + k) for k in range(STEPS)]
    ref = _episode(xy, th, acts)
    res = [torch.load(os.path.join(str(tmp_path), 'r%d.pt' % r)) for r in range(world)]
    for r in res:
        assert r['returns'].shape == (TOTAL_ENVS,)
        assert torch.equal(r['returns'], ref.episode_returns)          # gathered in global env order on every rank
        lo, hi = r['range']
        assert torch.equal(r['poses'], ref.sim.poses()[lo:hi])          # shard == rows of the unsharded run, bit for bit
    assert res[0]['range'] == (0, 3) and res[1]['range'] == (3, 5)
