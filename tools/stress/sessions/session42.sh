cd $GRAFT_REPO_ROOT
out=gpurun_out/r2q3; mkdir -p $out
export TMPDIR=/tmp
timeout 60 python -m pytest tests/test_gpu_fullsize.py -q -x -k "polynomial" 2>&1 | tail -4
(cd /tmp && timeout 80 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $GRAFT_REPO_ROOT/$out/bench_cfg2_under_rocprof.json 2> $GRAFT_REPO_ROOT/$out/bench_err)
python - <<PY
import csv, glob, json
d=json.load(open("$out/bench_cfg2_under_rocprof.json")); r=d["roofline"]
print("under rocprof:", d["value"], d["ms_per_step"], r["kernel"], r["avg_launch_ms"], r["alone_launch_ms"], r["frac"], "| exact:", d.get("exact_basis"), "| one:", d["one_render_in_flight"]["value"])
f = glob.glob("$out/prof/**/*kernel_stats.csv", recursive=True)[0]
for row in list(csv.reader(open(f)))[1:8]:
    print(row[0][:80].ljust(80), row[1], "%.1f us"%(float(row[3])/1e3))
PY
