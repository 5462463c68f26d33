# round 6, session 11: GPU suite + model step + quick line on the tree with the folded densify statistics
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/r06_s11_gpu_tests.log 2>&1; echo "pytest rc=$?"; tail -3 $O/r06_s11_gpu_tests.log
timeout 300 python tools/prof_model_step.py 30 2> $O/r06_s11_model_step.txt; cat $O/r06_s11_model_step.txt
timeout 300 python tools/bench_step.py 2>/dev/null | cut -c1-260; timeout 300 python tools/bench_step.py --graph 2>/dev/null | cut -c1-260
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-latency --no-other-configs 2>/dev/null > $O/r06_s11_bench_cfg2_quick.json
python - <<PY
import json
r=json.load(open("$O/r06_s11_bench_cfg2_quick.json")); h=r.get("heads_path") or {}
print("value", round(r["value"],1), "one-step", round((r.get("one_step_in_flight") or {}).get("value",0),1), "frac", r["roofline"].get("frac"), "| heads", round(h.get("value",0),1), "one-step", round((h.get("one_step_in_flight") or {}).get("value",0),1),
      "| autograd", round((r.get("autograd_surface") or {}).get("value",0),1), "model", round((r.get("model_surface") or {}).get("value",0),1), "dropin", round((r.get("dropin_gs_surface") or {}).get("value",0),1))
PY
