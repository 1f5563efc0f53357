"""Build libkilobots_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m gym_kilobots_amd.build [--force]
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = [os.path.join(HERE, 'csrc', 'kb_sim.hip')]
INC = os.path.join(ROOT, 'include')
LIB = os.path.join(HERE, 'libkilobots_hip.so')

# -ffp-contract=off: the step is specified as a sequence of individually rounded fp32 operations
# (DESIGN.md); fused multiply-adds would make results depend on compiler scheduling.
FLAGS = ['--offload-arch=gfx950', '-O3', '-ffp-contract=off', '-fno-fast-math', '-fPIC', '-shared', '-std=c++17']


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = SRC + [os.path.join(INC, 'kilobots_hip.h')]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get('HIPCC', 'hipcc')
    cmd = [hipcc] + FLAGS + ['-I', INC, '-o', LIB] + SRC
    if verbose:
        print(' '.join(cmd))
    subprocess.check_call(cmd)
    return LIB


def build_profile(verbose=False):
    """Diagnostic build with in-kernel cycle stamps per phase (tools/phase_profile.py); never shipped
    as the product library: it writes its stamps behind the status buffer."""
    hipcc = os.environ.get('HIPCC', 'hipcc')
    out = os.path.join(HERE, 'libkilobots_hip_prof.so')
    cmd = [hipcc] + FLAGS + ['-DKB_PROFILE', '-I', INC, '-o', out] + SRC
    if verbose:
        print(' '.join(cmd))
    subprocess.check_call(cmd)
    return out


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
