#!/bin/bash
# Copy the judged summaries of one measurement pass (gpurun_out/<tag>, made by tools/measure_round.sh) into profiles/
# under the round prefix: tools/refresh_profiles.sh <tag> <round prefix, e.g. r02>
set -e
R=gpurun_out/${1:?tag}; P=${2:?prefix}
cp $R/kernel_trace_summary.txt profiles/${P}_bench_kernel_trace_summary.txt
cp $R/kernel_stats.csv profiles/${P}_bench_kernel_stats.csv
cp $R/sq_counters.txt profiles/${P}_sq_counters.txt
cp $R/traffic_latest.json profiles/traffic_latest.json
cp $R/traffic_latest.json profiles/${P}_traffic.json
for f in bench_line bench_line_driver bench_cfg4_line bench_cfg4_boxes_line bench_cfg2_line bench_sense_line; do [ -f $R/$f.json ] && cp $R/$f.json profiles/${P}_$f.json; done
{ cat $R/phase_cycles.txt; echo "---- cfg4 ----"; cat $R/phase_cycles_cfg4.txt; } > profiles/${P}_phase_cycles.txt
python3 - <<PY
import json
R = '$R'
d = json.loads(open(R + '/bench_line.json').readline())
print('cfg3 %.3e frac %.4f traffic %.1f MB kernel %.1f us contacts/env %.0f fused %.3e cpu %.2e' % (d['value'], d['roofline']['frac'], (d['roofline']['traffic'] or 0) / 1e6, d['roofline']['avg_launch_ms'] * 1e3, d['contacts_per_env'], d['fused_env_step']['kilobot_steps_per_s_one_gpu'], d['cpu_baseline']['value']))
for f in ('line_driver', 'cfg4_line', 'cfg4_boxes_line', 'cfg2_line', 'sense_line'):
    try:
        print(f, '%.3e' % json.loads(open(R + '/bench_%s.json' % f).readline())['value'])
    except Exception as e:
        print(f, 'missing', e)
PY
