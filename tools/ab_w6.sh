cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/ab_w6
for lib in libkilobots_hip.so libkilobots_hip_w6.so; do
 for cfg in "--bots 480 --envs 8192 --arena 1.37 1.03" "--bots 1024 --envs 4096"; do
  KB_HIP_LIB=$GRAFT_REPO_ROOT/gym_kilobots_amd/$lib python3 bench.py --steps 40 --no-cpu-baseline --no-fused $cfg 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('%-26s %-45s %.4f ms  %.3e  contacts %.0f lds %d' % (sys.argv[1], sys.argv[2], d['roofline']['avg_launch_ms'], d['value'], d['contacts_per_env'], d['config']['lds_bytes_per_env']))" "$lib" "$cfg"
 done
done | tee gpurun_out/ab_w6/results.txt
