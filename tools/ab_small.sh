#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/ab_small
for lib in $1 libkilobots_hip.so; do
 for cfg in "--bots 16 --envs 65536" "--bots 40 --envs 32768" "--bots 64 --envs 16384" "--bots 100 --envs 16384" "--bots 128 --envs 16384" "--bots 200 --envs 16384" "--bots 256 --envs 16384" "--bots 64 --envs 256"; do
  KB_HIP_LIB=$GRAFT_REPO_ROOT/gym_kilobots_amd/$lib python3 bench.py --steps 40 --settle 40 --no-cpu-baseline --no-fused $cfg 2>/dev/null | python3 tools/ab_line.py "$lib $cfg"
 done
done | tee gpurun_out/ab_small/results.txt
