#!/bin/bash
# Copy the judged summaries of one measurement pass (gpurun_out/<tag>, made by tools/measure_round.sh) into profiles/.
set -e
R=gpurun_out/${1:?tag}
T=$(ls -t $R/trace/*/*kernel_trace.csv | head -1); S=$(ls -t $R/trace/*/*kernel_stats.csv | head -1)
python3 tools/summarise_trace.py $T > /tmp/kb_sum.txt
python3 - <<'PY'
new = open('/tmp/kb_sum.txt').read().split('\n')
p = 'profiles/r01_bench_kernel_trace_summary.txt'
out = []
for l in open(p).read().rstrip('\n').split('\n'):
    rep = [n for n in new if l[:18] and n.startswith(l[:18])]
    if l.startswith('grid '):
        n = [x for x in new if x.startswith('grid ')][0]
        l = l[:l.index('scratch ')] + 'scratch ' + n.split('scratch ')[1]
    elif rep and not l.startswith('rocprofv3') and not l.startswith('kernel:'):
        l = rep[0]
    out.append(l)
open(p, 'w').write('\n'.join(out) + '\n')
PY
cp $S profiles/r01_bench_kernel_stats.csv
for f in bench_line bench_cfg4_line bench_cfg4_boxes_line bench_cfg2_line; do cp $R/$f.json profiles/r01_$f.json; done
{ cat $R/phase_cycles.txt; echo "---- cfg4 ----"; cat $R/phase_cycles_cfg4.txt; } > profiles/r01_phase_cycles.txt
cp $R/traffic_latest.json profiles/traffic_latest.json
python3 - <<PY
import json
R = '$R'
d = json.loads(open(R + '/bench_line.json').readline())
print('cfg3 %.3e frac %.4f traffic %.1f MB kernel %.1f us fused %.3e cpu %.2e' % (d['value'], d['roofline']['frac'], d['roofline']['traffic'] / 1e6, d['roofline']['avg_launch_ms'] * 1e3, d['fused_env_step']['kilobot_steps_per_s_one_gpu'], d['cpu_baseline']['value']))
for f in ('cfg4', 'cfg4_boxes', 'cfg2'):
    print(f, '%.3e' % json.loads(open(R + '/bench_%s_line.json' % f).readline())['value'])
PY
