# round 6, session 5: GPU suite with the heads as separate images / in-kernel background and depth variance / view-parallel projection
# backward; the model-level step again (tools/prof_model_step.py) with its kernel trace; the heads and SH lines
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/r06_s5_gpu_tests.log 2>&1; echo "pytest rc=$?"; tail -3 $O/r06_s5_gpu_tests.log
timeout 300 python tools/prof_model_step.py 30 2> $O/r06_s5_model_step.txt; cat $O/r06_s5_model_step.txt
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-latency 2>/dev/null > $O/r06_s5_bench_cfg2_quick.json
python - <<PY
import json
r=json.load(open("$O/r06_s5_bench_cfg2_quick.json")); h=r.get("heads_path") or {}
print("value", round(r["value"],1), "one-step", round((r.get("one_step_in_flight") or {}).get("value",0),1), "frac", r["roofline"].get("frac"), "| heads", round(h.get("value",0),1), "one-step", round((h.get("one_step_in_flight") or {}).get("value",0),1),
      "| autograd", round((r.get("autograd_surface") or {}).get("value",0),1), "model", round((r.get("model_surface") or {}).get("value",0),1), "dropin", round((r.get("dropin_gs_surface") or {}).get("value",0),1))
PY
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r06_s5_prof -o prof -- python $R/tools/prof_model_step.py 30 > /dev/null 2> $O/r06_s5_prof.err
f=$(find $O/r06_s5_prof -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/r06_s5_model_step_kernel_stats.csv
rm -rf $O/r06_s5_prof
