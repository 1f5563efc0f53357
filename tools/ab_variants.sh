mkdir -p gpurun_out/ab1; cd $GRAFT_REPO_ROOT
for v in "" "--no-toi" "--vel-iters 1" "--vel-iters 1 --pos-iters 0 --no-toi" "--sense 0.065" "--sense 0.1"; do
  python3 bench.py --steps 40 --no-cpu-baseline --no-fused $v 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('%-45s %.4f ms  %.3e  contacts %.0f' % (sys.argv[1], d['roofline']['avg_launch_ms'], d['value'], d['contacts_per_env']))" "$v"
done | tee gpurun_out/ab1/results.txt
