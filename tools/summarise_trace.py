#!/usr/bin/env python3
"""Summary of a `rocprofv3 --kernel-trace --stats` run of bench.py: durations of the kb_step_kernel launches split into
warm-up, timed and fused launches.  usage: tools/summarise_trace.py <kernel_trace.csv> [warmup=60] [steps=200]"""
import csv
import sys


def main():
    path = sys.argv[1]
    warm = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 200
    rows = [r for r in csv.DictReader(open(path)) if 'kb_step_kernel' in r['Kernel_Name']]
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    dur = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in rows]
    r0 = rows[warm]
    print('rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --no-cpu-baseline  (MI355X)')
    print('kernel: %s' % r0['Kernel_Name'])
    print('grid %s x wg %s; VGPR %s (+%s AGPR) SGPR %s; dynamic LDS %s B; scratch %s B/lane'
          % (r0['Grid_Size_X'], r0['Workgroup_Size_X'], r0['VGPR_Count'], r0['Accum_VGPR_Count'], r0['SGPR_Count'],
             r0['LDS_Block_Size'], r0['Scratch_Size']))
    timed, fused = dur[warm:warm + steps], dur[warm + steps:]
    print('launches: %d (%d warm-up, %d timed 1-substep launches, then %d fused 10-substep launches)'
          % (len(dur), warm, len(timed), len(fused)))
    print('timed launches (index %d..%d): avg %.1f us, min %.1f us, max %.1f us'
          % (warm, warm + len(timed) - 1, sum(timed) / len(timed), min(timed), max(timed)))
    print('warm-up launch 0 (no contacts yet): %.1f us' % dur[0])
    if fused:
        print('fused 10-substep launches: avg %.1f us' % (sum(fused) / len(fused)))


if __name__ == '__main__':
    main()
