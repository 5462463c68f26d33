cd $GRAFT_REPO_ROOT
out=gpurun_out/r2h; mkdir -p $out
for a in "--slots 2" "--slots 3" "--slots 2 --batch 16" "--slots 4 --batch 4" "--slots 1"; do
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-latency $a > "$out/bench.json" 2> $out/bench.err
  python -c "
import json,sys
d=json.load(open(sys.argv[1])); r=d['roofline']
print(sys.argv[2], round(d['value'],1), r['fwd_kernel'], 'bwd', round(r['avg_launch_ms'],3), 'alone', round(r['alone_launch_ms'],3), 'fwd', round(r['fwd_launch_ms'],3), 'alone fwd', round(r['alone_fwd_launch_ms'],3), 'host us/step', round(d['timing']['host_enqueue_ms_per_step']*1e3))" "$out/bench.json" "$a" || tail -5 $out/bench.err
done
for c in cfg3 cfg4; do for a in "--batch 8" "--batch 4" "--batch 2 --slots 3"; do
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-latency --config $c $a > "$out/bench.json" 2> $out/bench.err
  python -c "
import json,sys
d=json.load(open(sys.argv[1])); r=d['roofline']
print(sys.argv[2], round(d['value'],1), 'bwd', round(r['avg_launch_ms'],3), 'alone', round(r['alone_launch_ms'],3), 'fwd', round(r['fwd_launch_ms'],3))" "$out/bench.json" "$c $a" || tail -5 $out/bench.err
done; done
