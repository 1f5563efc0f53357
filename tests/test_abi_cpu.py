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


def test_struct_layouts_match_header(tmp_path):
    """The ctypes mirrors (product binding and oracle binding) have the size and field offsets that a C compiler gives
    the structs of include/kilobots_hip.h and oracle/kb_oracle.h."""
    import subprocess
    from oracle import oracle as O
    fields_cfg = [n for n, _ in nat.KbConfig._fields_]
    fields_buf = list(nat.BUFFER_FIELDS)
    fields_rp = [n for n, _ in nat.KbResetParams._fields_]
    src = ['#include <stdio.h>', '#include <stddef.h>', '#include "kilobots_hip.h"', '#include "kb_oracle.h"', 'int main(void) {',
           'printf("%zu %zu %zu %zu %zu %zu\\n", sizeof(kb_config), sizeof(kb_buffers), sizeof(kb_reset_params), sizeof(kbo_config), sizeof(kbo_state), sizeof(kbo_reset_params));']
    for f in fields_cfg:
        src.append('printf("%%zu %%zu\\n", offsetof(kb_config, %s), offsetof(kbo_config, %s));' % (f, f))
    for f in fields_buf:
        src.append('printf("%%zu\\n", offsetof(kb_buffers, %s));' % f)
    for f in fields_rp:
        src.append('printf("%%zu %%zu\\n", offsetof(kb_reset_params, %s), offsetof(kbo_reset_params, %s));' % (f, f))
    for f, _ in O.State._fields_:
        src.append('printf("%%zu\\n", offsetof(kbo_state, %s));' % f)
    src.append('return 0; }')
    c = tmp_path / 'layout.c'
    c.write_text('\n'.join(src))
    exe = str(tmp_path / 'layout')
    subprocess.check_call(['gcc', '-I', os.path.join(ROOT, 'include'), '-I', os.path.join(ROOT, 'oracle'), str(c), '-o', exe])
    out = subprocess.check_output([exe]).decode().split('\n')
    sizes = [int(v) for v in out[0].split()]
    assert sizes == [C.sizeof(nat.KbConfig), C.sizeof(nat.KbBuffers), C.sizeof(nat.KbResetParams),
                     C.sizeof(O.Config), C.sizeof(O.State), C.sizeof(O.ResetParams)]
    assert [n for n, _ in O.Config._fields_] == fields_cfg          # the two configs share one layout
    k = 1
    for f in fields_cfg:
        a, b = (int(v) for v in out[k].split())
        assert a == getattr(nat.KbConfig, f).offset and b == getattr(O.Config, f).offset and a == b, f
        k += 1
    for f in fields_buf:
        assert int(out[k]) == getattr(nat.KbBuffers, f).offset, f
        k += 1
    for f in fields_rp:
        a, b = (int(v) for v in out[k].split())
        assert a == getattr(nat.KbResetParams, f).offset and b == getattr(O.ResetParams, f).offset, f
        k += 1
    for f, _ in O.State._fields_:
        assert int(out[k]) == getattr(O.State, f).offset, f
        k += 1


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


def test_lds_images_of_the_round_two_layouts(lib):
    """Residency is what decides throughput: the sorted-bin image of the kernels without objects (round 3;
    hashed bins for sparse swarms, staging entries traded for resident envs) keeps the sizes that hold 3 / 20 / 24+ envs per CU."""
    h = C.c_void_p()

    def lds(n, **kw):
        cfg = nat.default_config(4096, n, **kw)
        assert lib.kb_create(C.byref(cfg), C.byref(h)) == 0
        v = lib.kb_lds_bytes(h)
        lib.kb_destroy(h)
        return v
    assert lds(1024) <= 52 * 1024                      # three envs per CU (the hardware hands LDS out in granules: < 53 248)
    assert lds(1024, num_objects=4) <= 80 * 1024       # two
    assert lds(1024, contact_capacity=12000) <= 53760      # 42 LDS granules of 1280 B: three envs per CU like the fixed-size kernel
    assert lds(64) <= 8192 + 512 and lds(16) <= 5632   # hashed cell heads: 20 / 24+ one-wave envs per CU
    assert lds(512) <= 32 * 1024 and lds(768) <= 52 * 1024     # 512 kilobots: five 4-wave envs per CU (staging down to N + 64 entries)
    assert lds(400) <= 27 * 1024 and lds(64) <= 6400 + 256    # six envs; 24 one-wave envs
    # a small arena has fewer cells than the hash table would have entries: plain heads, still the compact image
    assert lds(100, world_width=0.6, world_height=0.6) <= lds(100) + 256


def test_sim_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from gym_kilobots_amd.sim import KilobotSim
    with pytest.raises(nat.KilobotsHipError):
        KilobotSim(1, 16)


def test_object_configuration_is_validated(lib):
    h = C.c_void_p()
    W = 25.0

    def create(**kw):
        cfg = nat.default_config(2, 16, **kw)
        rc = lib.kb_create(C.byref(cfg), C.byref(h))
        if rc == 0:
            lib.kb_destroy(h)
        return rc

    box = [[0.075 * W, 0.075 * W]]
    tri = [[1.25, -1.25], [1.25, 2.5], [-2.5, -1.25]]
    assert create(num_objects=1, obj_shape=[nat.SHAPE_BOX], obj_verts=[box]) == 0
    assert create(num_objects=1, obj_shape=[nat.SHAPE_POLYGON], obj_nverts=[3], obj_verts=[tri]) == 0
    # two fixtures on one body + a disc on another
    assert create(num_objects=2, num_fixtures=3, obj_fixture_body=[0, 1, 0], obj_shape=[nat.SHAPE_BOX, nat.SHAPE_CIRCLE, nat.SHAPE_POLYGON],
                  obj_nverts=[4, 0, 3], obj_radius=[0.0, 0.05, 0.0], obj_verts=[box, [[0, 0]], tri]) == 0
    bad = [
        dict(num_objects=1, obj_shape=[7]),                                                     # unknown shape
        dict(num_objects=1, obj_shape=[nat.SHAPE_BOX], obj_verts=[[[0.0, 1.0]]]),                # empty box
        dict(num_objects=1, obj_shape=[nat.SHAPE_POLYGON], obj_nverts=[2], obj_verts=[tri]),     # too few vertices
        dict(num_objects=1, obj_shape=[nat.SHAPE_POLYGON], obj_nverts=[3], obj_verts=[tri[::-1]]),   # clockwise
        dict(num_objects=1, obj_shape=[nat.SHAPE_CIRCLE], obj_radius=[0.0]),                     # no radius
        dict(num_objects=2, num_fixtures=1),                                                    # fewer fixtures than objects
        dict(num_objects=1, num_fixtures=2, obj_fixture_body=[0, 3], obj_shape=[nat.SHAPE_BOX] * 2, obj_verts=[box, box]),   # body out of range
        dict(num_objects=1, num_fixtures=2, obj_fixture_body=[0, 0], obj_shape=[nat.SHAPE_CIRCLE, nat.SHAPE_BOX],
             obj_radius=[0.05, 0.0], obj_verts=[[[0, 0]], box]),                                  # a circle shares its body
        dict(num_objects=2, num_fixtures=2, obj_fixture_body=[0, 0], obj_shape=[nat.SHAPE_BOX] * 2, obj_verts=[box, box]),   # object 1 has no fixture
        dict(num_objects=1, obj_shape=[nat.SHAPE_BOX], obj_verts=[box], obj_friction=-1.0),
    ]
    for kw in bad:
        assert create(**kw) == nat.KB_EINVAL, kw
        assert lib.kb_last_error()


def test_contact_capacity_matches_the_oracle(lib):
    from oracle import oracle as O
    h = C.c_void_p()
    for n in (1, 2, 7, 16, 33, 64, 100, 300, 512, 1000, 1024):
        for m in (0, 1, 4, 8):
            cfg = nat.default_config(3, n, num_objects=m)
            assert lib.kb_create(C.byref(cfg), C.byref(h)) == 0
            cap = lib.kb_contact_capacity(h)
            lds = lib.kb_lds_bytes(h)
            lib.kb_destroy(h)
            ocfg = O.default_config(3, n, num_objects=m)
            assert cap == O.lib().kbo_contact_capacity(C.byref(ocfg)), (n, m)
            assert lds <= 160 * 1024
            if n == 1024:
                assert lds <= 80 * 1024, 'two envs of 1024 kilobots must share a CU (LDS %d B with %d objects)' % (lds, m)
