#!/bin/bash
# A/B of experiment libraries on the headline workload + the tail probe (env counts that fill the CUs' slots evenly or not)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/ab2b
{
for lib in "$@"; do
  KB_HIP_LIB=$GRAFT_REPO_ROOT/gym_kilobots_amd/$lib python3 bench.py --steps 100 --no-cpu-baseline --no-fused 2>/dev/null | python3 tools/ab_line.py "$lib"
done
for n in 3072 3840 4096 4608; do
  python3 bench.py --steps 60 --no-cpu-baseline --no-fused --envs $n 2>/dev/null | python3 tools/ab_line.py "release lib, envs $n"
done
} | tee gpurun_out/ab2b/results.txt
