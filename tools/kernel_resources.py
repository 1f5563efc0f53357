#!/usr/bin/env python3
"""Per-kernel register / scratch / spill table of the kernel units (hipcc -Rpass-analysis=kernel-resource-usage).

usage: tools/kernel_resources.py [unit ...]      e.g. tools/kernel_resources.py d0 d0_discs      (default: all kb_inst_* units)
Prints one line per kernel: VGPRs, SGPRs, scratch bytes per lane, waves per SIMD, spilled SGPRs / VGPRs, and the number of
"Folded Spill" stores in the generated code."""
import glob
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gym_kilobots_amd import build as B  # noqa: E402


def unit_table(src, extra=()):
    out = os.path.join('/tmp', 'kbres_' + os.path.basename(src)[:-4] + '.s')
    cmd = [os.environ.get('HIPCC', 'hipcc')] + B.FLAGS + list(extra) + ['-I', B.INC, '-I', B.CSRC, '--cuda-device-only', '-S', src, '-o', out,
                                                        '-Rpass-analysis=kernel-resource-usage']
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if p.returncode:
        raise RuntimeError(p.stdout.decode()[-3000:])
    rows, cur = [], None
    for line in p.stdout.decode().split('\n'):
        m = re.search(r'remark:\s+(Function Name|TotalSGPRs|VGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]): (\S+)', line)
        if not m:
            continue
        k, v = m.group(1), m.group(2)
        if k == 'Function Name':
            cur = {'name': v}
            rows.append(cur)
        elif cur is not None:
            cur[k.split(' [')[0]] = v
    asm = open(out).read()
    for r in rows:
        i = asm.find('\n' + r['name'] + ':')
        j = asm.find('.end_amdhsa_kernel', i) if i >= 0 else -1
        body = asm[i:j] if i >= 0 else ''
        r['folded_spill_stores'] = len(re.findall(r'scratch_store[^\n]*Folded Spill', body))
    return rows


def demangle(n):
    try:
        return subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-cxxfilt', n], stdout=subprocess.PIPE).stdout.decode().strip()
    except Exception:
        return n


def main():
    sel = sys.argv[1:]
    units = [s for s in B.sources() if os.path.basename(s).startswith('kb_inst_')]
    if sel:
        units = [u for u in units if any(os.path.basename(u)[:-4] == 'kb_inst_' + s for s in sel)]
    with ThreadPoolExecutor(max_workers=min(8, len(units))) as ex:
        tabs = list(ex.map(unit_table, units))
    print('%-78s %5s %5s %8s %5s %7s %7s %7s' % ('kernel', 'VGPR', 'SGPR', 'scratch', 'w/SIMD', 'sgprSp', 'vgprSp', 'spillSt'))
    for u, rows in zip(units, tabs):
        for r in rows:
            if 'kb_step_kernel' not in r['name']:
                continue
            n = demangle(r['name']).replace('void kb::', '').replace('(kb::Params)', '')
            print('%-78s %5s %5s %8s %5s %7s %7s %7d' % (os.path.basename(u)[8:-4] + ':' + n, r.get('VGPRs'), r.get('TotalSGPRs'), r.get('ScratchSize'),
                                                      r.get('Occupancy'), r.get('SGPRs Spill'), r.get('VGPRs Spill'), r['folded_spill_stores']))


if __name__ == '__main__':
    main()
