cd $GRAFT_REPO_ROOT
out=gpurun_out/r2z; mkdir -p $out
export TMPDIR=/tmp
echo "== gpu tests (full, HEAD)"
( time timeout 420 python -m pytest tests -m gpu -q ) > $out/pytest.log 2>&1; tail -6 $out/pytest.log
echo "== smoke"
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
echo "== bench (driver command)"
timeout 240 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_cfg2.json 2> $out/bench_err; python - <<PY
import json
d=json.load(open("$out/bench_cfg2.json")); r=d["roofline"]; o=d["one_render_in_flight"]
print(d["value"], d["ms_per_step"], r["frac"], "bwd", r["avg_launch_ms"], "alone", r["alone_launch_ms"], r.get("alone_valu_frac"), "| one in flight", o["value"], o.get("hipgraph_replay"), "| cpu", d["cpu_baseline"]["value"])
PY
echo "== bench under rocprof kernel trace"
(cd /tmp && timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $GRAFT_REPO_ROOT/$out/bench_cfg2_under_rocprof.json 2>> $GRAFT_REPO_ROOT/$out/bench_err)
find $out/prof -name "*kernel_stats.csv" | head -3
python - <<PY
import csv, glob, json
f = glob.glob("$out/prof/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.reader(open(f)))[1:16]:
    if 'at::' in r[0]: continue
    print(r[0][:70].ljust(70), r[1], "%.1f us"%(float(r[3])/1e3))
d=json.load(open("$out/bench_cfg2_under_rocprof.json")); print("under rocprof:", d["value"], d["roofline"]["avg_launch_ms"], d["roofline"]["alone_launch_ms"])
PY
