cd $GRAFT_REPO_ROOT
out=gpurun_out/r2g; mkdir -p $out
for v in "X=1" "GSGEN_PPL_FWD_BATCH=2 GSGEN_PPL_FWD=2" "GSGEN_PPL_FWD_BATCH=4 GSGEN_PPL_FWD=4" "X=2" "GSGEN_PPL_FWD_BATCH=2 GSGEN_PPL_FWD=2"; do
  env $v timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > "$out/bench.json" 2> $out/bench.err
  python -c "
import json,sys
d=json.load(open(sys.argv[1])); r=d['roofline']; o=d.get('one_render_in_flight',{})
print(sys.argv[2][-40:], round(d['value'],1), r['fwd_kernel'], 'bwd', round(r['avg_launch_ms'],3), 'alone', round(r['alone_launch_ms'],3), 'fwd', round(r['fwd_launch_ms'],3), 'alone fwd', round(r['alone_fwd_launch_ms'],3), 'one', round(o.get('value',0),1), 'one fwd', round(o.get('fwd_kernel_ms',0),4))" "$out/bench.json" "$v" || tail -5 $out/bench.err
done
