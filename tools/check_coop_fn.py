#!/usr/bin/env python3
"""Quick bit-exactness check of the cooperative (giant-island) solver on the fixed-size kernel: works with -DKB_ONLY_BENCH
experiment libraries (KB_HIP_LIB=...).  Dense 1024-kilobot swarms, solver_mode 2 (LDS staging) and 4 (global staging)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O
from tests import scenes
from tests.test_parity_gpu import make_pair, assert_same, assert_ws_same, dev

for mode in (2, 4, 0):
    for pitch in (0.040, 0.034):
        E, N = 2, 1024
        xy, th = scenes.lattice_spawn(E, N, seed=3, pitch=pitch)
        osim, gsim = make_pair(E, N, xy=xy, th=th, solver_mode=mode)
        for k in range(6):
            a = scenes.random_actions(E, N, seed=40 + k)
            osim.set_actions(a)
            n = 1 if k < 4 else 3
            osim.step(n)
            gsim.step(n, actions=dev(a))
            assert_same(osim, gsim, 'mode %d pitch %.3f step %d' % (mode, pitch, k))
            assert_ws_same(osim, gsim, 'mode %d step %d' % (mode, k))
        print('mode', mode, 'pitch', pitch, 'ok: contacts', osim.count_contacts(0), 'status', osim.status, gsim.status.cpu().numpy())
