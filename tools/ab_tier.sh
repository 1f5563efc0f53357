#!/bin/bash
# register tiers (128 / 80 VGPRs) of the kernels without objects, size by size: KB_TIER=0|2 forces one
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/ab_tier
for t in 0 2; do
 for cfg in "--bots 16 --envs 65536" "--bots 40 --envs 32768" "--bots 64 --envs 16384" "--bots 100 --envs 16384" "--bots 128 --envs 16384" "--bots 200 --envs 16384" "--bots 256 --envs 16384" "--bots 400 --envs 8192" "--bots 512 --envs 8192" "--bots 768 --envs 4096"; do
  KB_TIER=$t python3 bench.py --steps 40 --settle 40 --no-cpu-baseline --no-fused $cfg 2>/dev/null | python3 tools/ab_line.py "tier $t $cfg"
 done
done | tee gpurun_out/ab_tier/results.txt
