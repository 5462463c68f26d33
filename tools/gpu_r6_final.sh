# round 6, final evidence on the final tree: the whole GPU suite, then tools/gpu_final.sh (driver line, traces, PMC passes, cfg3 / cfg4, stress lines),
# then the one-rank distributed line beside the plain one
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -m gpu -q --maxfail=12 -p no:cacheprovider > $O/r06_gpu_tests.log 2>&1; echo "pytest rc=$?" >> $O/r06_gpu_tests.log; tail -3 $O/r06_gpu_tests.log
bash tools/gpu_final.sh r06 2>&1 | tail -70
cd $R
for r in 1 2; do
  GSGEN_BENCH_FORCE_DIST=1 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > $O/r06_bench_cfg2_force_dist_$r.json 2>/dev/null
  python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > $O/r06_bench_cfg2_plain_same_box_$r.json 2>/dev/null
done
python - <<'PY' | tee $O/r06_force_dist_agreement.txt
import json
for r in (1, 2):
    a = json.load(open(f"gpurun_out/r06_bench_cfg2_force_dist_{r}.json")); b = json.load(open(f"gpurun_out/r06_bench_cfg2_plain_same_box_{r}.json"))
    print(f"round {r}: GSGEN_BENCH_FORCE_DIST=1 value {a['value']:.1f} (without the gather {a['value_no_gather']:.1f}) | plain {b['value']:.1f} | ratio {a['value']/b['value']:.4f} (without the gather {a['value_no_gather']/b['value']:.4f})")
PY
