cd $GRAFT_REPO_ROOT
out=gpurun_out/r2q4; mkdir -p $out
export TMPDIR=/tmp
timeout 60 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_cfg2.json 2> $out/bench_err; python - <<PY
import json
d=json.load(open("$out/bench_cfg2.json")); r=d["roofline"]
print(d["value"], d["ms_per_step"], r["kernel"], r["frac"], "bwd", r["avg_launch_ms"], "alone", r["alone_launch_ms"], "| exact", d.get("exact_basis", {}).get("value"), "| one", d["one_render_in_flight"]["value"], "| cpu", d["cpu_baseline"]["value"])
PY
GSGEN_BENCH_FORCE_DIST=1 timeout 40 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-latency 2> $out/dist_err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('force-dist', d['value'], d['config']['gather'][:30], d['config']['sh_basis'][:40])" || tail -3 $out/dist_err
timeout 30 python -m pytest tests/test_gpu_api.py -x -q 2>&1 | tail -2
