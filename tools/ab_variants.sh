#!/bin/bash
# A/B helper (GPU box): bench.py variants of the headline workload, one line each: tools/ab_variants.sh [variant ...]
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/ab1
if [ $# -eq 0 ]; then set -- "" "--no-toi" "--vel-iters 1" "--vel-iters 1 --pos-iters 0 --no-toi" "--sense 0.065" "--sense 0.1"; fi
for v in "$@"; do
  python3 bench.py --steps 40 --no-cpu-baseline --no-fused $v 2>/dev/null | python3 tools/ab_line.py "$v"
done | tee gpurun_out/ab1/results.txt
