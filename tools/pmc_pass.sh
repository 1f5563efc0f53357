#!/bin/bash
# One extra rocprofv3 counter pass over the bench workload: tools/pmc_pass.sh <tag> <counter> [<counter> ...]
# (program directly after `--`; counters in their own run with --kernel-trace only)
set -e -o pipefail
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/pmc -- python3 $ROOT/bench.py --steps 20 --no-cpu-baseline > $OUT/pmc.log 2>&1 || { tail -5 $OUT/pmc.log; exit 1; }
python3 $ROOT/tools/collect_counters.py 20 $(ls $OUT/pmc/*/*counter_collection.csv) | tee $OUT/counters.txt
find $OUT -name "*.csv" -size +8M -delete; find $OUT -name "*.db" -delete
