cd $GRAFT_REPO_ROOT
out=gpurun_out/r2o; mkdir -p $out
export TMPDIR=/tmp
echo "== gpu tests"
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
echo "== bench (driver command) cfg2"
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_cfg2.json 2> $out/bench_err; python - <<PY
import json
d=json.load(open("$out/bench_cfg2.json")); r=d["roofline"]; o=d["one_render_in_flight"]
print(d["value"], d["ms_per_step"], r["kernel"], "alone", r["alone_launch_ms"], "fwd alone", r["alone_fwd_launch_ms"], "| one in flight", o["value"], o.get("hipgraph_replay"))
PY
for c in cfg1 cfg3 cfg4; do
  timeout 600 python bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_$c.json 2>> $out/bench_err
  python -c "import json; d=json.load(open('$out/bench_$c.json')); print('$c', d['value'], d['ms_per_step'], d['config'].get('cameras_per_step'), d['config'].get('steps_in_flight'))"
done
echo "== PMC passes"
bash tools/pmc.sh fetch "FETCH_SIZE" 2>&1 | tail -3
bash tools/pmc.sh write "WRITE_SIZE" 2>&1 | tail -3
bash tools/pmc.sh sq "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" 2>&1 | tail -3
cp gpurun_out/pmc_fetch.json gpurun_out/pmc_write.json gpurun_out/pmc_sq.json $out/ 2>/dev/null
echo "== kernel stats"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $GRAFT_REPO_ROOT/$out/bench_cfg2_under_rocprof.json 2>> $GRAFT_REPO_ROOT/$out/bench_err)
head -12 $out/prof/bench_kernel_stats.csv | cut -c1-160
