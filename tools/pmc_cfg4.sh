#!/bin/bash
# SQ counter passes + HBM traffic of the cfg4 workloads (discs / boxes): tools/pmc_cfg4.sh <tag>
set -e -o pipefail
TAG=${1:-cfg4pmc}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/$TAG
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in discs boxes; do
  X=""; [ $v = boxes ] && X="--boxes"
  B="python3 $ROOT/bench.py --steps 20 --no-cpu-baseline --no-fused --objects 4 $X"
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --kernel-trace --output-format csv -d $OUT/${v}_a -- $B > $OUT/${v}_a.log 2>&1 || echo "a failed"
  rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT --kernel-trace --output-format csv -d $OUT/${v}_b -- $B > $OUT/${v}_b.log 2>&1 || echo "b failed"
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/${v}_f -- $B > $OUT/${v}_f.log 2>&1 || echo "f failed"
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/${v}_w -- $B > $OUT/${v}_w.log 2>&1 || echo "w failed"
  python3 $ROOT/tools/collect_counters.py 20 $(ls $OUT/${v}_*/*/*counter_collection.csv 2>/dev/null) > $OUT/${v}_counters.txt 2>&1 || true
  echo $v done
done
find $OUT -name "*.csv" -size +2M -delete; find $OUT -name "*.db" -delete
cat $OUT/discs_counters.txt $OUT/boxes_counters.txt
