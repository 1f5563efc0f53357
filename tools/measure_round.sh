#!/bin/bash
# Full measurement pass of one round on the GPU box (run through gpurun from the repo root):
#   bench line (with CPU baseline), rocprofv3 kernel trace, FETCH_SIZE / WRITE_SIZE PMC passes (separate runs, as the
#   microarch guide prescribes), cfg4 lines, phase profile.  Everything lands in gpurun_out/$1/.
set -e -o pipefail
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/$TAG
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python3 $ROOT/bench.py --steps 20 --no-cpu-baseline > $OUT/pmc_fetch.log 2>&1
echo fetch done
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- python3 $ROOT/bench.py --steps 20 --no-cpu-baseline > $OUT/pmc_write.log 2>&1
echo write done
F=$(ls $OUT/fetch/*/*counter_collection.csv | head -1); W=$(ls $OUT/write/*/*counter_collection.csv | head -1)
python3 $ROOT/tools/collect_traffic.py $F $W 4096 1024 $ROOT/profiles/traffic_latest.json > $OUT/traffic.json
cp $ROOT/profiles/traffic_latest.json $OUT/traffic_latest.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --no-cpu-baseline > $OUT/trace_bench.json 2> $OUT/trace_bench.err
echo trace done
cd $ROOT
python3 bench.py > $OUT/bench_line.json 2> $OUT/bench.err
echo bench done
python3 bench.py --steps 40 --no-cpu-baseline --objects 4 > $OUT/bench_cfg4_line.json 2>> $OUT/bench.err
python3 bench.py --steps 40 --no-cpu-baseline --objects 4 --boxes > $OUT/bench_cfg4_boxes_line.json 2>> $OUT/bench.err
python3 bench.py --envs 256 --bots 64 --steps 400 --warmup 100 --no-cpu-baseline > $OUT/bench_cfg2_line.json 2>> $OUT/bench.err
KB_HIP_LIB=$ROOT/gym_kilobots_amd/libkilobots_hip_prof.so python3 tools/phase_profile.py > $OUT/phase_cycles.txt 2>> $OUT/bench.err
KB_HIP_LIB=$ROOT/gym_kilobots_amd/libkilobots_hip_prof.so python3 tools/phase_profile.py --objects 4 > $OUT/phase_cycles_cfg4.txt 2>> $OUT/bench.err
# keep the merge small: the raw traces stay on the box except the per-kernel csv files
find $OUT -name "*.csv" -size +8M -delete
echo all done
