cd $GRAFT_REPO_ROOT
out=gpurun_out/r2j; mkdir -p $out
timeout 2400 python -m pytest tests -m gpu -q -s > $out/pytest.log 2>&1; echo "pytest rc=$? $(tail -1 $out/pytest.log)"
grep -n "^FAILED\|^ERROR\|host cost per" $out/pytest.log | head -20
grep -n "Error\|assert " $out/pytest.log | head -30
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver.json 2> $out/bench.err; python -c "
import json
d=json.load(open('$out/bench_driver.json')); r=d['roofline']; o=d.get('one_render_in_flight',{})
print(round(d['value'],1), d['config']['cameras_per_step'], r['kernel'], r['fwd_kernel'], 'bwd', round(r['avg_launch_ms'],3), 'alone', round(r['alone_launch_ms'],3), '| one', round(o.get('value',0),1), d['timing']['host_enqueue_ms_per_step'])" || tail -5 $out/bench.err
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
