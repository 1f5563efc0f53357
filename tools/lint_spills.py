#!/usr/bin/env python3
"""Static check of the generated gfx950 code for one shape of a register-allocator bug met in round 1 (ROCm 7.2, clang 22):

A VGPR spill store of a value that is live in ALL lanes is placed at the top of the join block of a divergent region,
BEFORE the `s_or_b64 exec, exec, <saved>` that widens the EXEC mask again:

    .LBB7_2697:
        scratch_store_dwordx2 off, v[26:27], off offset:124   ; 8-byte Folded Spill     <- EXEC still = lanes of the `if`
        s_or_b64 exec, exec, s[6:7]                                                     <- ... widened only here
        ...
        scratch_load_dwordx2 v[2:3], off, off offset:124      ; 8-byte Folded Reload    <- under the full mask

Lanes that did not take the branch are never written to the slot, the reload hands them stale scratch memory.  In the build
that showed it the spilled values were the polygon manifold of register slot 0 (stored under the mask of slot 1); a polygon
contact in slot 0 on a lane outside that mask then got a wrong position correction ("Robustness").

usage: tools/lint_spills.py file.s [...]      (hipcc --cuda-device-only -S);  exit status 1 if the shape is found
"""
import re
import sys


def lint(path):
    hits = []
    kernel, block = None, []

    def flush():
        for k, (idx, line) in enumerate(block):
            if 'Folded Spill' not in line or not line.strip().startswith('scratch_store'):
                continue
            for _, later in block[k + 1:]:
                code = later.split(';')[0]
                if re.search(r's_or_b64 exec, exec', code):
                    hits.append((kernel, idx + 1, line.strip()))
                    break
                if re.search(r'saveexec|s_mov_b64 exec|s_and_b64 exec|s_andn2_b64 exec', code):
                    break

    for idx, line in enumerate(open(path).read().split('\n')):
        m = re.match(r'^(_Z\w+):', line)
        if m or re.match(r'^\.LBB\d+_\d+:', line):
            flush()
            block = []
            if m:
                kernel = m.group(1)
        block.append((idx, line))
    flush()
    return hits


def main():
    bad = 0
    for path in sys.argv[1:]:
        for kernel, lineno, text in lint(path):
            print('%s:%d: %s: spill store ahead of the EXEC restore of its block: %s' % (path, lineno, kernel, text))
            bad += 1
    print('%d suspicious spill store(s)' % bad)
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
