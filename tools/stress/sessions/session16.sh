cd $GRAFT_REPO_ROOT
out=gpurun_out/r2p; mkdir -p $out
export TMPDIR=/tmp
echo "== gpu tests"
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
echo "== bench (driver command) cfg2"
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_cfg2.json 2> $out/bench_err; python - <<PY
import json
d=json.load(open("$out/bench_cfg2.json")); r=d["roofline"]; o=d["one_render_in_flight"]
print(d["value"], d["ms_per_step"], r["kernel"], "alone", r["alone_launch_ms"], "fwd alone", r["alone_fwd_launch_ms"], "| one in flight", o["value"], o.get("hipgraph_replay"), o["fwd_kernel_ms"], o["bwd_kernel_ms"])
PY
for c in cfg3 cfg4; do
  timeout 600 python bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline --no-latency > $out/bench_$c.json 2>> $out/bench_err
  python -c "import json; d=json.load(open('$out/bench_$c.json')); print('$c', d['value'], d['ms_per_step'])"
done
bash tools/pmc.sh sq "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" 2>&1 | tail -3
