cd $GRAFT_REPO_ROOT
out=gpurun_out/r2z2; mkdir -p $out
export TMPDIR=/tmp
echo "== model-level golden through the fused path"
timeout 300 python -m pytest tests/test_gpu_golden.py -q -x 2>&1 | tail -15
echo "== optimisation step (cfg5 without the diffusion model)"
timeout 200 python tools/bench_step.py > $out/step_eager.json 2> $out/step_err; cat $out/step_eager.json
timeout 200 python tools/bench_step.py --graph > $out/step_graph.json 2>> $out/step_err; cat $out/step_graph.json
tail -3 $out/step_err
echo "== kernel trace of the eager step"
(cd /tmp && timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/prof -o step -- python $GRAFT_REPO_ROOT/tools/bench_step.py --steps 50 --warmup 10 > /dev/null 2>> $GRAFT_REPO_ROOT/$out/step_err)
python - <<PY
import csv, glob
f = glob.glob("$out/prof/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.reader(open(f)))[1:]
tot = sum(int(r[1]) for r in rows); tt = sum(float(r[2]) for r in rows)
print("launches per step ~", tot / 61.0, "kernel time per step us ~", tt / 61.0 / 1e3)
for r in rows[:40]:
    print(r[0][:90].ljust(90), r[1], "%.1f us" % (float(r[3]) / 1e3), r[4])
PY
