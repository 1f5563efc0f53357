#!/bin/bash
# sims with the sleep state at the two register budgets of the kernels without objects (KB_TIER=0|2), size by size
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/ab_sleep_tier
for t in 0 2; do
 for cfg in "--bots 16 --envs 65536" "--bots 64 --envs 16384" "--bots 128 --envs 16384" "--bots 256 --envs 16384" "--bots 512 --envs 8192" "--bots 1024 --envs 4096"; do
  KB_TIER=$t python3 bench.py --steps 40 --settle 40 --no-cpu-baseline --no-fused --sleep $cfg 2>/dev/null | python3 tools/ab_line.py "sleep tier $t $cfg"
 done
done | tee gpurun_out/ab_sleep_tier/results.txt
