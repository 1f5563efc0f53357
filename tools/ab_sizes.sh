#!/bin/bash
# size sweep of the generic kernels, release library against an experiment library: tools/ab_sizes.sh <variant lib name>
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/ab_sizes
for lib in $1 libkilobots_hip.so; do
 for cfg in "--bots 64 --envs 16384" "--bots 128 --envs 16384" "--bots 256 --envs 16384" "--bots 400 --envs 8192" "--bots 512 --envs 8192" "--bots 768 --envs 4096"; do
  KB_HIP_LIB=$GRAFT_REPO_ROOT/gym_kilobots_amd/$lib python3 bench.py --steps 40 --settle 40 --no-cpu-baseline --no-fused $cfg 2>/dev/null | python3 tools/ab_line.py "$lib $cfg"
 done
done | tee gpurun_out/ab_sizes/results.txt
