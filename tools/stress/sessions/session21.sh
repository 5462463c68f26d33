cd $GRAFT_REPO_ROOT
out=gpurun_out/r2u; mkdir -p $out
export TMPDIR=/tmp
echo "== gpu tests"
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
echo "== bench cfg2 under rocprof kernel trace"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $GRAFT_REPO_ROOT/$out/bench_cfg2_under_rocprof.json 2>> $GRAFT_REPO_ROOT/$out/bench_err)
python - <<PY
import csv
for r in list(csv.reader(open("$out/prof/bench_kernel_stats.csv")))[1:32]:
    if 'at::' in r[0]: continue
    print(r[0][:64].ljust(64), r[1], "%.1f us"%(float(r[3])/1e3))
PY
echo "== bench cfg2"
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_cfg2.json 2> $out/bench_err; python - <<PY
import json
d=json.load(open("$out/bench_cfg2.json")); r=d["roofline"]; o=d["one_render_in_flight"]
print(d["value"], d["ms_per_step"], "alone", r["alone_launch_ms"], "fwd alone", r["alone_fwd_launch_ms"], "| one in flight", o["value"], o["fwd_kernel_ms"], o["bwd_kernel_ms"], o.get("hipgraph_replay"))
PY
