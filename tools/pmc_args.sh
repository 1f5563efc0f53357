#!/bin/bash
# rocprofv3 counter pass over bench.py with extra bench arguments: tools/pmc_args.sh <tag> "<bench args>" <counter> [...]
set -e -o pipefail
TAG=$1; shift
ARGS=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/pmc -- python3 $ROOT/bench.py --steps 20 --no-cpu-baseline $ARGS > $OUT/pmc.log 2>&1 || { tail -5 $OUT/pmc.log; exit 1; }
python3 $ROOT/tools/collect_counters.py 20 $(ls $OUT/pmc/*/*counter_collection.csv) | tee $OUT/counters.txt
rm -rf $OUT/pmc
