#!/usr/bin/env python3
"""Average rocprofv3 --pmc counters over the timed kb_step_kernel launches of a bench.py run.
usage: tools/collect_counters.py <counter_collection.csv> [steps=20] [more csv files ...]
Prints one line per counter: mean per launch over the timed launches (the `steps` launches in front of the fused tail)."""
import csv
import sys
from collections import defaultdict


def main():
    paths = [a for a in sys.argv[1:] if a.endswith('.csv')]
    steps = [int(a) for a in sys.argv[1:] if a.isdigit()]
    steps = steps[0] if steps else 20
    nf = max(2, min(10, steps // 10)) + 1
    for path in paths:
        per = defaultdict(list)
        name = None
        for r in csv.DictReader(open(path)):
            if 'kb_step_kernel' not in r['Kernel_Name']:
                continue
            if name is not None and r['Kernel_Name'] != name:
                continue                      # the headline instantiation only (the sleep-state leg launches another one)
            name = r['Kernel_Name']
            per[r['Counter_Name']].append((int(r.get('Dispatch_Id', 0) or 0), float(r['Counter_Value'])))
        print('# %s' % path.split('/')[-1])
        print('# kernel: %s' % name)
        for c in sorted(per):
            v = [x for _, x in sorted(per[c])]
            t = v[-(2 * steps + nf):-(steps + nf)]      # the timed region (behind it: its audit pass, then the fused launches)
            print('%-28s %16.1f   (mean of %d timed launches; launch 0: %.1f)' % (c, sum(t) / len(t), len(t), v[0]))


if __name__ == '__main__':
    main()
