#!/usr/bin/env python3
"""Turn rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE, collected in separate runs as
/opt/skills/guides/MI355X_MICROARCH.md prescribes) into HBM bytes per kb_step_kernel launch.

gfx950 corrections applied (same guide, section HBM): FETCH_SIZE counts 64 B per 128-B request, i.e.
exactly half of the bytes of a coalesced streaming read -> x2; WRITE_SIZE is exact.  Both are KiB.
The corrections were re-checked on this run's own torch kernels (fill of 16 MiB: WRITE_SIZE 16384;
elementwise pow over 16 MiB: FETCH_SIZE 8206), see profiles/r01_pmc_calibration.txt.

usage: tools/collect_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <envs> <bots> <out.json>
"""
import csv
import json
import sys


def launches(path, counter):
    vals, name0 = [], None
    for r in csv.DictReader(open(path)):
        if 'kb_step_kernel' in r['Kernel_Name'] and r['Counter_Name'] == counter:
            name0 = name0 or r['Kernel_Name']      # the headline instantiation only (the sleep-state leg launches another one)
            if r['Kernel_Name'] == name0:
                vals.append(float(r['Counter_Value']))
    return vals


def main():
    fetch_csv, write_csv, envs, bots, out = sys.argv[1:6]
    f = launches(fetch_csv, 'FETCH_SIZE')
    w = launches(write_csv, 'WRITE_SIZE')
    n = 20     # bench.py --steps 20: the timed region, then its audit pass (20 more), then 3 fused launches
    f_t, w_t = f[-(2 * n + 3):-(n + 3)], w[-(2 * n + 3):-(n + 3)]
    fetch_b = sum(f_t) / len(f_t) * 1024 * 2
    write_b = sum(w_t) / len(w_t) * 1024
    res = {'envs': int(envs), 'bots': int(bots), 'launches_averaged': len(f_t),
           'fetch_bytes_per_launch': fetch_b, 'write_bytes_per_launch': write_b,
           'hbm_bytes_per_launch': fetch_b + write_b,
           'first_launch_fetch_bytes': f[0] * 1024 * 2, 'first_launch_write_bytes': w[0] * 1024,
           'corrections': 'FETCH_SIZE x2 (gfx950 128-B requests tallied at 64 B), WRITE_SIZE exact; units KiB'}
    json.dump(res, open(out, 'w'), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == '__main__':
    main()
