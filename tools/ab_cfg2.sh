#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/ab_cfg2
for lib in libkilobots_hip_prev.so libkilobots_hip.so; do
 for cfg in "--bots 64 --envs 256" "--bots 64 --envs 2048" "--bots 16 --envs 65536" "--bots 1024 --envs 4096 --objects 4" ; do
  KB_HIP_LIB=$GRAFT_REPO_ROOT/gym_kilobots_amd/$lib python3 bench.py --steps 200 --settle 40 --no-cpu-baseline --no-fused $cfg 2>/dev/null | python3 tools/ab_line.py "$lib $cfg"
 done
done | tee gpurun_out/ab_cfg2/results.txt
