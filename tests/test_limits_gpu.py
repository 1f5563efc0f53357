"""The staging limits that only the device step has (63 partners of one kilobot in one direction, 255 contacts in one
(cell, direction) group, 64 kilobots on one fixture): an env that runs into one is FLAGGED (status bit 2) -- and every other env
of the same launch is still bit-exact against the oracle, which has no such limits."""
import numpy as np
import pytest
import torch

from oracle import oracle as O
from tests import scenes
from tests.test_parity_gpu import make_pair, cpu, dev

pytestmark = pytest.mark.gpu

XB, Y0 = -1.0 + 0.035 * 28, -0.75 + 0.035 * 21 + 0.0175      # a cell boundary in x, the middle of a cell row in y (metres)


def _scene(E, N, env0):
    xy, th = scenes.lattice_spawn(E, N, seed=5, pitch=0.045)
    far = xy[0].copy()
    k = len(env0)
    xy[0, :k] = env0
    # the rest of env 0: the far half of its lattice (nothing near the pile)
    keep = far[np.abs(far[:, 0] - XB) + np.abs(far[:, 1] - Y0) > 0.2]
    xy[0, k:] = keep[:N - k]
    return xy, th


def _check(osim, gsim, steps=3, fields=('x', 'y', 'theta')):
    E = osim.x.shape[0]
    for k in range(steps):
        a = scenes.random_actions(E, osim.x.shape[1], seed=90 + k)
        osim.set_actions(a)
        osim.step(1)
        gsim.step(1, actions=dev(a))
        torch.cuda.synchronize()
        for f in fields:
            assert np.array_equal(getattr(osim, f)[1:], cpu(getattr(gsim, f)).reshape(getattr(osim, f).shape)[1:]), (k, f)
    sg = cpu(gsim.status)
    assert sg[0] & 4, 'env 0 ran into the limit and must say so: status %s' % sg
    assert ((sg[1:] & ~2) == 0).all() and ((osim.status[1:] & 1) == 0).all(), (sg, osim.status)      # the others: nothing dropped


def test_more_than_63_partners_in_one_direction_is_flagged_and_the_other_envs_stay_exact():
    E, N = 3, 128
    pile = [[XB - 0.001, Y0]] + [[XB + 0.001 + 0.00005 * i, Y0 + 0.004 * np.sin(i)] for i in range(66)]
    xy, th = _scene(E, N, np.array(pile))
    osim, gsim = make_pair(E, N, xy=xy, th=th)
    _check(osim, gsim)


def test_more_than_255_contacts_in_one_cell_pair_is_flagged_and_the_other_envs_stay_exact():
    E, N = 3, 128
    west = [[XB - 0.001 - 0.0002 * i, Y0 + 0.003 * np.cos(i)] for i in range(20)]
    east = [[XB + 0.001 + 0.0002 * i, Y0 + 0.003 * np.sin(i)] for i in range(20)]
    xy, th = _scene(E, N, np.array(west + east))
    osim, gsim = make_pair(E, N, xy=xy, th=th, contact_capacity=4096)
    _check(osim, gsim)


def test_more_than_64_kilobots_on_one_fixture_is_flagged_and_the_other_envs_stay_exact():
    E, N = 3, 128
    ring = [[XB + 0.05 * np.cos(0.09 * i), Y0 + 0.05 * np.sin(0.09 * i)] for i in range(70)]     # 70 kilobots inside a disc of 0.075 m
    xy, th = _scene(E, N, np.array(ring))
    objs = np.tile(np.array([[XB, Y0]])[None], (E, 1, 1))
    objs[1:] = [[0.6, 0.4]]
    osim, gsim = make_pair(E, N, xy=xy, th=th, objects=objs, obj_radius=[0.075] * 8, contact_capacity=4096)
    _check(osim, gsim, fields=('x', 'y', 'theta', 'ox', 'oy', 'otheta'))
