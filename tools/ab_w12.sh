cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/ab_w12
run() { KB_HIP_LIB=$GRAFT_REPO_ROOT/gym_kilobots_amd/$1 python3 bench.py --steps 40 --no-cpu-baseline --no-fused $2 2>&1 | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('%-26s %-30s %.4f ms  %.3e  contacts %.0f lds %d thr %d status %d' % (sys.argv[1], sys.argv[2], d['roofline']['avg_launch_ms'], d['value'], d['contacts_per_env'], d['config']['lds_bytes_per_env'], d['config']['workgroup_threads'], d['status_flags']))" "$1" "$2"; }
run libkilobots_hip.so ""
run libkilobots_hip_w12.so "--threads 768"
run libkilobots_hip_w10.so "--threads 640"
run libkilobots_hip_w6.so ""
