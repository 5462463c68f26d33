cd $GRAFT_REPO_ROOT
out=gpurun_out/r2i; mkdir -p $out
for a in "--config cfg2" "--config cfg2 --latency-segments 1" "--config cfg2 --latency-segments 4" "--config cfg3" "--config cfg4" "--config cfg1"; do
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline $a > "$out/bench.json" 2> $out/bench.err
  python -c "
import json,sys
d=json.load(open(sys.argv[1])); r=d['roofline']; o=d.get('one_render_in_flight',{})
print(sys.argv[2], round(d['value'],1), 'B', d['config']['cameras_per_step'], 'slots', d['config']['steps_in_flight'], 'bwd', round(r['avg_launch_ms'],3), 'alone', round(r['alone_launch_ms'],3), 'fwd', round(r['fwd_launch_ms'],3), '| one', round(o.get('value',0),1), 'fwd', round(o.get('fwd_kernel_ms',0),4), 'bwd', round(o.get('bwd_kernel_ms',0),4), 'graph', round(o.get('hipgraph_replay',{}).get('value',0),1))" "$out/bench.json" "$a" || tail -5 $out/bench.err
done
