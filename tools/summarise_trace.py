#!/usr/bin/env python3
"""Summary of a `rocprofv3 --kernel-trace --stats` run of bench.py: durations of the kb_step_kernel launches split into
settle + warm-up, timed and fused launches (counted from the END of the run: bench.py's pre-roll length is adaptive).
usage: tools/summarise_trace.py <kernel_trace.csv> [steps=200] [label]"""
import csv
import sys


def main():
    path = sys.argv[1]
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    label = sys.argv[3] if len(sys.argv) > 3 else 'python3 bench.py --no-cpu-baseline'
    rows = [r for r in csv.DictReader(open(path)) if 'kb_step_kernel' in r['Kernel_Name']]
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    # the headline instantiation only (bench.py's extra leg with the sleep state launches another one between the timed and the fused launches)
    rows = [r for r in rows if r['Kernel_Name'] == rows[0]['Kernel_Name']]
    dur = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in rows]
    nf = max(2, min(10, steps // 10)) + 1          # bench.py: one untimed + nf timed fused launches at the end
    pre = len(dur) - 2 * steps - nf                # ... behind the timed region and its audit pass (the same launches with per-launch events)
    r0 = rows[pre]
    print('rocprofv3 --kernel-trace --stats --output-format csv -- %s  (MI355X)' % label)
    print('kernel: %s' % r0['Kernel_Name'])
    print('grid %s x wg %s; VGPR %s (+%s AGPR) SGPR %s; LDS column %s B (static only; dynamic LDS = kb_lds_bytes()); scratch %s B/lane'
          % (r0['Grid_Size_X'], r0['Workgroup_Size_X'], r0['VGPR_Count'], r0['Accum_VGPR_Count'], r0['SGPR_Count'],
             r0['LDS_Block_Size'], r0['Scratch_Size']))
    timed, audit, fused = dur[pre:pre + steps], dur[pre + steps:pre + 2 * steps], dur[pre + 2 * steps:]
    print('launches: %d (%d settle + warm-up, %d timed 1-substep launches, %d audit launches, then %d fused 10-substep launches)'
          % (len(dur), pre, len(timed), len(audit), len(fused)))
    print('audit launches (per-launch events): avg %.1f us' % (sum(audit) / max(1, len(audit))))
    print('timed launches (index %d..%d): avg %.1f us, min %.1f us, max %.1f us, first %.1f us, last %.1f us'
          % (pre, pre + len(timed) - 1, sum(timed) / len(timed), min(timed), max(timed), timed[0], timed[-1]))
    print('launch 0 (no contacts yet): %.1f us;  launches 5..24 (contact build-up): avg %.1f us;  last 20 of the pre-roll: avg %.1f us'
          % (dur[0], sum(dur[5:25]) / 20, sum(dur[pre - 20:pre]) / 20))
    if fused:
        print('fused 10-substep launches: avg %.1f us' % (sum(fused[1:]) / max(1, len(fused) - 1)))


if __name__ == '__main__':
    main()
