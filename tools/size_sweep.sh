#!/bin/bash
# kilobot-steps/s of the release library over swarm sizes (settled scenes, one substep per launch, sleeping off / on):
#   tools/size_sweep.sh > gpurun_out/size_sweep.txt      (GPU box)
cd ${GRAFT_REPO_ROOT:-$PWD}
for cfg in "--bots 16 --envs 65536" "--bots 40 --envs 32768" "--bots 64 --envs 16384" "--bots 100 --envs 16384" "--bots 128 --envs 16384" "--bots 256 --envs 16384" "--bots 400 --envs 8192" "--bots 512 --envs 8192" "--bots 768 --envs 4096" "--bots 1024 --envs 4096"; do
 for sl in "" "--sleep"; do
  python3 bench.py --steps 40 --settle 40 --no-cpu-baseline --no-fused $sl $cfg 2>/dev/null | python3 tools/ab_line.py "release $cfg $sl"
 done
done
