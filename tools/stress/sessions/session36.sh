cd $GRAFT_REPO_ROOT
out=gpurun_out/r2p1; mkdir -p $out
export TMPDIR=/tmp
for a in "--geo-priority 1" "--geo-priority 0" "--geo-priority 1 --slots 3" "--geo-priority 1"; do
  timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-latency $a > "$out/bench.json" 2> $out/bench.err
  python -c "
import json,sys
d=json.load(open(sys.argv[1])); r=d['roofline']
print(sys.argv[2], '|', round(d['value'],1), 'ms/step', round(d['ms_per_step'],3), 'bwd', round(r['avg_launch_ms'],3), 'alone', round(r['alone_launch_ms'],3), 'fwd', round(r['fwd_launch_ms'],3), 'alone fwd', round(r['alone_fwd_launch_ms'],3), 'host us/step', round(d['timing']['host_enqueue_ms_per_step']*1e3), d['timing']['renders_per_s_min'], d['timing']['renders_per_s_max'])" "$out/bench.json" "$a" || tail -5 $out/bench.err
done
