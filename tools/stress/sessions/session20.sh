cd $GRAFT_REPO_ROOT
out=gpurun_out/r2t; mkdir -p $out
export TMPDIR=/tmp
echo "== gpu tests"
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
echo "== bench default (driver command)"
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_cfg2.json 2> $out/bench_err; python - <<PY
import json
d=json.load(open("$out/bench_cfg2.json")); r=d["roofline"]; o=d["one_render_in_flight"]
print(d["value"], d["ms_per_step"], r["frac"], r["traffic"], "alone", r["alone_launch_ms"], r.get("alone_valu_frac"), "| one in flight", o["value"], "| cpu", d["cpu_baseline"]["value"])
print(open("$out/bench_cfg2.json").read().count("\n"), "lines on stdout")
PY
for c in cfg3 cfg4; do
  timeout 600 python bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_$c.json 2>> $out/bench_err
  python -c "import json; d=json.load(open('$out/bench_$c.json')); print('$c', d['value'], d['ms_per_step'], d['config'].get('cameras_per_step'), d['config'].get('steps_in_flight'), d['one_render_in_flight']['value'])"
done
