#!/bin/bash
# Full measurement pass of one round on the GPU box (run through gpurun from the repo root):
#   bench line (with CPU legs), rocprofv3 kernel trace, FETCH_SIZE / WRITE_SIZE PMC passes (separate runs, as the
#   microarch guide prescribes), SQ / LDS counter passes, cfg4 / cfg2 / sensing lines, phase profile.
#   Everything lands in gpurun_out/$1/.
set -e -o pipefail
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/$TAG
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $ROOT/bench.py --steps 20 --no-cpu-baseline"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- $B > $OUT/pmc_fetch.log 2>&1
echo fetch done
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- $B > $OUT/pmc_write.log 2>&1
echo write done
F=$(ls $OUT/fetch/*/*counter_collection.csv | head -1); W=$(ls $OUT/write/*/*counter_collection.csv | head -1)
python3 $ROOT/tools/collect_traffic.py $F $W 4096 1024 $ROOT/profiles/traffic_latest.json > $OUT/traffic.json
cp $ROOT/profiles/traffic_latest.json $OUT/traffic_latest.json
# SQ counters (8 slots per pass): where the waves' cycles go, and what the LDS pipeline does
rocprofv3 -L > $OUT/counters_available.txt 2>&1 || true
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --kernel-trace --output-format csv -d $OUT/sq_a -- $B > $OUT/pmc_sq_a.log 2>&1 || echo "sq_a pass failed"
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM --kernel-trace --output-format csv -d $OUT/sq_b -- $B > $OUT/pmc_sq_b.log 2>&1 || echo "sq_b pass failed"
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_FLAT GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/sq_c -- $B > $OUT/pmc_sq_c.log 2>&1 || echo "sq_c pass failed"
python3 $ROOT/tools/collect_counters.py 20 $(ls $OUT/sq_*/*/*counter_collection.csv 2>/dev/null) > $OUT/sq_counters.txt 2>&1 || true
echo counters done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --steps 200 --no-cpu-baseline > $OUT/trace_bench.json 2> $OUT/trace_bench.err
python3 $ROOT/tools/summarise_trace.py $(ls -t $OUT/trace/*/*kernel_trace.csv | head -1) 200 "python3 bench.py --steps 200 --no-cpu-baseline" > $OUT/kernel_trace_summary.txt
cp $(ls -t $OUT/trace/*/*kernel_stats.csv | head -1) $OUT/kernel_stats.csv
echo trace done
cd $ROOT
python3 bench.py --steps 20 --warmup 5 > $OUT/bench_line_driver.json 2> $OUT/bench.err
python3 bench.py > $OUT/bench_line.json 2>> $OUT/bench.err
echo bench done
python3 bench.py --steps 40 --no-cpu-baseline --objects 4 > $OUT/bench_cfg4_line.json 2>> $OUT/bench.err
python3 bench.py --steps 40 --no-cpu-baseline --objects 4 --boxes > $OUT/bench_cfg4_boxes_line.json 2>> $OUT/bench.err
python3 bench.py --envs 256 --bots 64 --steps 400 --no-cpu-baseline > $OUT/bench_cfg2_line.json 2>> $OUT/bench.err
python3 bench.py --steps 40 --no-cpu-baseline --sense 0.07 > $OUT/bench_sense_line.json 2>> $OUT/bench.err
KB_HIP_LIB=$ROOT/gym_kilobots_amd/libkilobots_hip_prof.so python3 tools/phase_profile.py > $OUT/phase_cycles.txt 2>> $OUT/bench.err
KB_HIP_LIB=$ROOT/gym_kilobots_amd/libkilobots_hip_prof.so python3 tools/phase_profile.py --objects 4 > $OUT/phase_cycles_cfg4.txt 2>> $OUT/bench.err
# keep the merge small: the raw traces stay on the box except the per-kernel csv files
find $OUT -name "*.csv" -size +8M -delete
find $OUT -name "*.db" -delete
echo all done
