#!/bin/bash
# the headline workload on several experiment libraries: tools/ab_libs.sh lib1.so lib2.so ...
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/ab_libs
for lib in "$@"; do
  KB_HIP_LIB=$GRAFT_REPO_ROOT/gym_kilobots_amd/$lib python3 bench.py --steps 60 --no-cpu-baseline --no-fused 2>/dev/null | python3 tools/ab_line.py "$lib"
done | tee gpurun_out/ab_libs/results.txt
