cd $GRAFT_REPO_ROOT
out=gpurun_out/r2c; mkdir -p $out
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc=$? $(tail -1 $out/pytest.log)"
grep -n "FAILED\|Error\|assert" $out/pytest.log | head -20
echo "== A/B bench"
for v in "GSGEN_BWD_SH_PACKED=1" "GSGEN_BWD_SH_PACKED=0" "GSGEN_PPL_BWD_SH_BATCH=2" "GSGEN_BWD_MFMA_BATCH=2"; do
  env $v timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_$v.json 2> $out/bench_$v.err
  python -c "
import json
d=json.load(open('$out/bench_$v.json')); r=d['roofline']; o=d.get('one_render_in_flight',{})
print('$v', round(d['value'],1), r['kernel'], 'bwd', round(r['avg_launch_ms'],3), 'alone', round(r['alone_launch_ms'],3), 'fwd', round(r['fwd_launch_ms'],3), 'one', round(o.get('value',0),1), 'graph', o.get('hipgraph_replay',{}).get('value'))" || tail -5 $out/bench_$v.err
done
for c in cfg3 cfg4 cfg1; do
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-latency --config $c > $out/bench_$c.json 2> $out/bench_$c.err
  python -c "
import json
d=json.load(open('$out/bench_$c.json')); r=d['roofline']
print('$c', round(d['value'],1), r['kernel'], 'bwd', round(r['avg_launch_ms'],3), 'alone', round(r['alone_launch_ms'],3), 'fwd', round(r['fwd_launch_ms'],3))" || tail -5 $out/bench_$c.err
done
