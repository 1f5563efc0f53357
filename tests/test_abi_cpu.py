"""C-ABI checks that need no GPU: the library builds, loads and exports every symbol that
include/kilobots_hip.h declares; host-side argument validation and error reporting."""
import ctypes as C
import os
import re

import pytest

from gym_kilobots_amd import _native as nat
from gym_kilobots_amd import build as kb_build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def lib():
    kb_build.build()
    return nat.load()


def test_exports_every_declared_symbol(lib):
    hdr = open(os.path.join(ROOT, 'include', 'kilobots_hip.h')).read()
    declared = set(re.findall(r'\b(kb_[a-z_]+)\s*\(', hdr))
    assert declared == set(nat.EXPORTS)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.kb_version().startswith(b'kilobots_hip')


def test_struct_sizes_match_header():
    # 3 ints, 2+1 floats, 2 ints, 2 ints, 4+1 floats, 8 floats, 1 float, 1 int, 8 floats, 4 floats, 1 int
    # ... + solver_mode, light_count, light_kind[4], 2 x [4], 4 x [4][2]
    assert C.sizeof(nat.KbConfig) == 4 * (3 + 3 + 2 + 2 + 5 + 8 + 1 + 1 + 8 + 4 + 1 + 1 + 1 + 4 + 8 + 32 + 8 + 8 + 64 + 1 + 1 + 8)
    assert C.sizeof(nat.KbBuffers) == 8 * len(nat.BUFFER_FIELDS)


def test_create_validates_and_reports(lib):
    h = C.c_void_p()
    cfg = nat.default_config(4, 64)
    assert lib.kb_create(C.byref(cfg), C.byref(h)) == 0
    assert 0 < lib.kb_lds_bytes(h) <= 160 * 1024
    assert lib.kb_contact_capacity(h) >= 4 * 64
    assert lib.kb_block_threads(h) % 64 == 0
    # stepping before kb_bind is an error, not a crash
    assert lib.kb_step(h, None, None, 1, 0, None) == nat.KB_ENOTBOUND
    assert b'kb_bind' in lib.kb_last_error()
    assert lib.kb_set_block_threads(h, 100) == nat.KB_EINVAL
    lib.kb_destroy(h)
    for bad in (dict(num_bots=0), dict(num_bots=nat.MAX_BOTS + 1), dict(num_envs=0), dict(dt=0.0),
                dict(num_objects=9), dict(ws_slots=0)):
        cfg = nat.default_config(1, 16)
        for k, v in bad.items():
            setattr(cfg, k, v)
        assert lib.kb_create(C.byref(cfg), C.byref(h)) == nat.KB_EINVAL, bad
        assert lib.kb_last_error()
    cfg = nat.default_config(1, 16, nat.DRIVE_PHOTOTAXIS, nat.LIGHT_NONE)
    assert lib.kb_create(C.byref(cfg), C.byref(h)) == nat.KB_EINVAL


def test_largest_config_fits_lds(lib):
    h = C.c_void_p()
    cfg = nat.default_config(4096, 1024)
    assert lib.kb_create(C.byref(cfg), C.byref(h)) == 0
    assert lib.kb_lds_bytes(h) <= 160 * 1024
    lib.kb_destroy(h)


def test_sim_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from gym_kilobots_amd.sim import KilobotSim
    with pytest.raises(nat.KilobotsHipError):
        KilobotSim(1, 16)
