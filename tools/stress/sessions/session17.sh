cd $GRAFT_REPO_ROOT
out=gpurun_out/r2q; mkdir -p $out
export TMPDIR=/tmp
echo "== gpu tests"
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
echo "== bench (driver command) cfg2 under rocprof kernel trace"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $GRAFT_REPO_ROOT/$out/bench_cfg2_under_rocprof.json 2>> $GRAFT_REPO_ROOT/$out/bench_err)
python - <<PY
import csv
for r in list(csv.reader(open("$out/prof/bench_kernel_stats.csv")))[1:30]:
    if 'at::' in r[0]: continue
    print(r[0][:64].ljust(64), r[1], "%.1f us"%(float(r[3])/1e3))
PY
echo "== bench cfg2"
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_cfg2.json 2> $out/bench_err; python - <<PY
import json
d=json.load(open("$out/bench_cfg2.json")); r=d["roofline"]; o=d["one_render_in_flight"]
print(d["value"], d["ms_per_step"], "alone", r["alone_launch_ms"], "fwd alone", r["alone_fwd_launch_ms"], "| one in flight", o["value"], o.get("hipgraph_replay"))
PY
for c in cfg1 cfg3 cfg4; do
  timeout 600 python bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_$c.json 2>> $out/bench_err
  python -c "import json; d=json.load(open('$out/bench_$c.json')); print('$c', d['value'], d['ms_per_step'], d['one_render_in_flight']['value'])"
done
echo "== public autograd paths"
for cfg in "--res 512 --batch 4" "--res 512 --batch 8" "--res 800 --batch 8" "--res 512 --batch 4 --heads"; do
  timeout 300 python tools/bench_batch.py --no-stats --steps 200 $cfg 2>&1 | tail -1
done
timeout 300 python tools/bench_frame.py 2>&1 | tail -1
