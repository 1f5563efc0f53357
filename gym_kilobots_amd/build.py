"""Build libkilobots_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m gym_kilobots_amd.build [--force]

The kernel template is instantiated per drive law in separate translation units
(csrc/kb_inst_d*.hip), compiled in parallel and linked with the C-ABI unit (csrc/kb_abi.hip).
"""
import glob
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, 'csrc')
INC = os.path.join(ROOT, 'include')
LIB = os.path.join(HERE, 'libkilobots_hip.so')
PROF_LIB = os.path.join(HERE, 'libkilobots_hip_prof.so')

# -ffp-contract=off: the step is specified as a sequence of individually rounded fp32 operations
# (DESIGN.md); fused multiply-adds would make results depend on compiler scheduling.
EXPECTED_HIPCC = 'HIP version: 7.2'     # toolchain of the round-1/2 parity and fuzz runs
FLAGS = ['--offload-arch=gfx950', '-O3', '-ffp-contract=off', '-fno-fast-math', '-fPIC', '-std=c++17']


def sources():
    return sorted(glob.glob(os.path.join(CSRC, '*.hip')))


def needs_build(lib=LIB):
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    deps = sources() + glob.glob(os.path.join(CSRC, '*.h')) + glob.glob(os.path.join(CSRC, '*.inc')) + [os.path.join(INC, 'kilobots_hip.h')]
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(args):
    src, obj, extra, verbose = args
    # -save-temps=obj keeps the gfx950 assembly of exactly the object that gets linked next to it: the codegen lint
    # (tools/lint_spills.py) reads that file
    cmd = [os.environ.get('HIPCC', 'hipcc')] + FLAGS + extra + ['-I', INC, '-I', CSRC, '-save-temps=obj', '-c', src, '-o', obj]
    if verbose:
        print(' '.join(cmd))
    if verbose:
        subprocess.check_call(cmd)
    else:       # keep the compiler's diagnostics: a failing build (also the automatic one on import) must say why
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        if r.returncode != 0:
            raise RuntimeError('hipcc failed (%d) on %s:\n%s' % (r.returncode, os.path.basename(src), r.stderr.decode(errors='replace')[-4000:]))
    # of the saved temporaries only the device assembly is wanted (the rest is hundreds of MB that would travel with
    # every gpurun snapshot)
    stem = obj[:-2]
    for f in glob.glob(stem + '-*') + glob.glob(stem + '.hip-*'):
        if not f.endswith('-hip-amdgcn-amd-amdhsa-gfx950.s'):
            os.remove(f)
    return obj


def _device_asm(objdir, src):
    return os.path.join(objdir, os.path.basename(src)[:-4] + '-hip-amdgcn-amd-amdhsa-gfx950.s')


def hipcc_version():
    try:
        out = subprocess.run([os.environ.get('HIPCC', 'hipcc'), '--version'], stdout=subprocess.PIPE, stderr=subprocess.STDOUT).stdout.decode()
        return ' | '.join(l.strip() for l in out.split('\n') if 'HIP version' in l or 'clang version' in l)
    except Exception as err:      # noqa: BLE001
        return 'unknown (%s)' % err


def _lint_asm(paths):
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import lint_spills
    findings = []
    for a in paths:
        findings += lint_spills.lint(a)
    return findings


def _build(lib, extra, tag, verbose, lint=True):
    objdir = os.path.join(HERE, '_obj', tag)
    os.makedirs(objdir, exist_ok=True)
    jobs = [(s, os.path.join(objdir, os.path.basename(s)[:-4] + '.o'), extra, verbose) for s in sources()]
    with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
        objs = list(ex.map(_compile, jobs))
    info = {'hipcc': hipcc_version(), 'flags': FLAGS + list(extra), 'lint': 'skipped'}
    if lint:
        # The register allocator of ROCm 7.2 (clang 22) was caught placing VGPR spill stores under a narrowed EXEC mask
        # (wrong physics, DESIGN.md "Robustness").  Every build is checked for that code shape and FAILS on a finding:
        # a different hipcc must not produce a silently miscompiled library.
        asm = [_device_asm(objdir, s_) for s_ in sources() if os.path.basename(s_).startswith('kb_inst_')]
        missing = [a for a in asm if not os.path.exists(a)]
        if missing:
            raise RuntimeError('codegen lint: hipcc did not leave the device assembly (%s); refusing to link an unchecked library' % missing[0])
        findings = _lint_asm(asm)
        if findings:
            if os.path.exists(lib):
                os.remove(lib)
            raise RuntimeError('codegen lint failed with %s: %d VGPR spill store(s) ahead of the EXEC restore of their block, e.g. %s'
                               % (info['hipcc'], len(findings), findings[0]))
        info['lint'] = 'clean (%d kernel units)' % len(asm)
        if EXPECTED_HIPCC not in info['hipcc']:
            print('WARNING: built with "%s"; the parity suite and the 20 000-scene fuzz were run with %s. '
                  'Run `python -m pytest tests -m gpu` and tools/debug_fuzz_seed.py sweeps before trusting this build.'
                  % (info['hipcc'], EXPECTED_HIPCC), file=sys.stderr)
    cmd = [os.environ.get('HIPCC', 'hipcc'), '--offload-arch=gfx950', '-shared', '-fPIC', '-o', lib] + objs
    if verbose:
        print(' '.join(cmd))
    subprocess.check_call(cmd)
    import json
    json.dump(info, open(os.path.join(objdir, 'build_info.json'), 'w'), indent=1)
    return lib


def lint_codegen(verbose=False):
    """Compile the kernel units to gfx950 assembly and run tools/lint_spills.py over them: the register allocator of
    ROCm 7.2 was caught placing VGPR spill stores under a narrowed EXEC mask (wrong results, see DESIGN.md
    "Robustness").  Returns the list of findings (empty = clean).  (build() runs the same lint on the assembly of the
    objects it links and fails on a finding; this entry point re-derives the assembly from the sources.)"""
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import lint_spills
    # the assembly that build() kept next to the objects it linked is the shipped code: use it while it is current
    reldir = os.path.join(HERE, '_obj', 'rel')
    kept = [_device_asm(reldir, s_) for s_ in sources() if os.path.basename(s_).startswith('kb_inst_')]
    deps = sources() + glob.glob(os.path.join(CSRC, '*.h')) + glob.glob(os.path.join(CSRC, '*.inc')) + [os.path.join(INC, 'kilobots_hip.h')]
    if os.path.exists(LIB) and all(os.path.exists(a) and os.path.getmtime(a) >= max(os.path.getmtime(d) for d in deps) for a in kept):
        findings = []
        for a in kept:
            findings += lint_spills.lint(a)
        return findings
    asmdir = os.path.join(HERE, '_obj', 'asm')
    os.makedirs(asmdir, exist_ok=True)
    units = [s_ for s_ in sources() if os.path.basename(s_).startswith('kb_inst_')]

    def emit(src):
        out = os.path.join(asmdir, os.path.basename(src)[:-4] + '.s')
        cmd = [os.environ.get('HIPCC', 'hipcc')] + FLAGS + ['-I', INC, '-I', CSRC, '--cuda-device-only', '-S', src, '-o', out]
        if verbose:
            print(' '.join(cmd))
        subprocess.check_call(cmd, stderr=subprocess.DEVNULL)
        return out
    with ThreadPoolExecutor(max_workers=min(8, len(units))) as ex:
        asm = list(ex.map(emit, units))
    findings = []
    for a in asm:
        findings += lint_spills.lint(a)
    return findings


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    return _build(LIB, [], 'rel', verbose)


def build_profile(verbose=False):
    """Diagnostic build with in-kernel cycle stamps per phase (tools/phase_profile.py); never shipped
    as the product library: it writes its stamps behind the status buffer."""
    return _build(PROF_LIB, ['-DKB_PROFILE'], 'prof', verbose, lint=False)


def build_variant(name, defines, only=None, verbose=False, standalone=False):
    """Experiment build: libkilobots_hip_<name>.so with extra -D flags (select it with KB_HIP_LIB=...).
    only: restrict the drive-law units that are compiled (e.g. ['d0']) to save time; the others come from the release build."""
    lib = os.path.join(HERE, 'libkilobots_hip_%s.so' % name)
    objdir = os.path.join(HERE, '_obj', name)
    os.makedirs(objdir, exist_ok=True)
    jobs, reuse = [], []
    for s in sources():
        base = os.path.basename(s)[:-4]
        if only is not None and base != 'kb_abi' and not any(base.endswith(o) for o in only):
            if not standalone:      # (standalone: the listed units define everything the C ABI references, e.g. -DKB_ONLY_BENCH)
                reuse.append(os.path.join(HERE, '_obj', 'rel', base + '.o'))
            continue
        jobs.append((s, os.path.join(objdir, base + '.o'), list(defines), verbose))
    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        objs = list(ex.map(_compile, jobs))
    subprocess.check_call([os.environ.get('HIPCC', 'hipcc'), '--offload-arch=gfx950', '-shared', '-fPIC', '-o', lib] + objs + reuse)
    return lib


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
