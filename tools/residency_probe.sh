#!/bin/bash
# phase cycles of wave 0 (diagnostic build) at 1, 2, 3 envs per CU and at the full grid: which phases stretch when envs share a CU
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/residency
for e in 256 512 768 4096; do
  echo "==== $e envs"; KB_HIP_LIB=$GRAFT_REPO_ROOT/gym_kilobots_amd/libkilobots_hip_prof.so python3 tools/phase_profile.py --envs $e --warm 110 --steps 20 2>/dev/null
done | tee gpurun_out/residency/results.txt
