"""Generated-code check (no GPU needed): the gfx950 assembly of every kernel unit is scanned for the one shape of a
register-allocator bug that produced wrong results during round 1 (a VGPR spill store placed ahead of the EXEC restore of
its block; tools/lint_spills.py, DESIGN.md "Robustness").  A finding means: perturb the code (or the compiler got fixed and
the lint is obsolete) -- never ship the build."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))


def test_lint_recognises_the_bad_shape(tmp_path):
    import lint_spills
    bad = tmp_path / 'bad.s'
    bad.write_text('\n'.join([
        '_Z6kernelv:', '\ts_and_saveexec_b64 s[6:7], vcc', '\ts_cbranch_execz .LBB0_2', '.LBB0_1:', '\tv_mov_b32_e32 v1, 0',
        '.LBB0_2:', '\tscratch_store_dwordx2 off, v[26:27], off offset:124 ; 8-byte Folded Spill',
        '\ts_or_b64 exec, exec, s[6:7]', '\tscratch_load_dwordx2 v[2:3], off, off offset:124 ; 8-byte Folded Reload', '\ts_endpgm']))
    good = tmp_path / 'good.s'
    good.write_text('\n'.join([
        '_Z6kernelv:', '\tscratch_store_dword off, v1, off offset:8 ; 4-byte Folded Spill', '\ts_and_saveexec_b64 s[6:7], vcc',
        '.LBB0_2:', '\ts_or_b64 exec, exec, s[6:7]', '\tscratch_store_dword off, v2, off offset:12 ; 4-byte Folded Spill', '\ts_endpgm']))
    assert len(lint_spills.lint(str(bad))) == 1
    assert lint_spills.lint(str(good)) == []


def test_shipped_kernels_have_no_spill_under_a_narrowed_exec_mask():
    from gym_kilobots_amd import build
    findings = build.lint_codegen()
    assert findings == [], '\n'.join('%s line %d: %s' % f for f in findings)
