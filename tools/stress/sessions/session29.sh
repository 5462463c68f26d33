cd $GRAFT_REPO_ROOT
out=gpurun_out/r2final; mkdir -p $out
export TMPDIR=/tmp
echo "== gpu tests (full)"
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
echo "== bench (driver command)"
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_cfg2.json 2> $out/bench_err; python - <<PY
import json
d=json.load(open("$out/bench_cfg2.json")); r=d["roofline"]; o=d["one_render_in_flight"]
print(d["value"], d["ms_per_step"], r["frac"], "alone", r["alone_launch_ms"], r.get("alone_valu_frac"), "| one in flight", o["value"], o.get("hipgraph_replay"), "| cpu", d["cpu_baseline"]["value"])
PY
for c in cfg1 cfg3 cfg4; do
  timeout 600 python bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_$c.json 2>> $out/bench_err
  python -c "import json; d=json.load(open('$out/bench_$c.json')); print('$c', d['value'], d['ms_per_step'], d['config'].get('cameras_per_step'), d['config'].get('steps_in_flight'), d['one_render_in_flight']['value'])"
done
echo "== one in flight with the packed per-camera forward (A/B)"
GSGEN_PPL_FWD=2 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); o=d['one_render_in_flight']; print('PPL_FWD=2', o['value'], o['fwd_kernel'], o['fwd_kernel_ms'])"
echo "== default python bench.py wall time"
( time timeout 900 python bench.py > $out/bench_default.json 2>/dev/null ) 2>&1 | grep real
python -c "import json; d=json.load(open('$out/bench_default.json')); print(d['value'], d['steps'], d['warmup'])"
