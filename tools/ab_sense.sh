#!/bin/bash
# neighbour sensing cost on the headline workload: release library against an experiment library (tools/ab_sense.sh lib.so)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/ab_sense
for lib in libkilobots_hip.so "$@"; do
 for r in 0 0.065 0.07 0.1; do
  X=""; [ $r != 0 ] && X="--sense $r"
  KB_HIP_LIB=$GRAFT_REPO_ROOT/gym_kilobots_amd/$lib python3 bench.py --steps 60 --no-cpu-baseline --no-fused $X 2>/dev/null | python3 tools/ab_line.py "$lib sense $r"
 done
done | tee gpurun_out/ab_sense/results.txt
