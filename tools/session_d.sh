cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py tests/test_gpu_golden.py -m gpu -q --maxfail=5 -p no:cacheprovider -k "not fuzz" > gpurun_out/r3d_gpu_tests.log 2>&1; tail -3 gpurun_out/r3d_gpu_tests.log
bash tools/ab.sh "--slots 3" "--slots 2" "--slots 4" "--slots 3 --variant ppl_fwd_batch=4" "--slots 3 --batch 4" "--slots 4 --batch 4" "--slots 3 --batch 12"
cp gpurun_out/ab.log gpurun_out/r3d_ab.log
for c in cfg3 cfg4 cfg1; do timeout 300 python bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r3d_bench_$c.json 2> gpurun_out/r3d_bench_$c.err; python - <<PY
import json
r=json.load(open("gpurun_out/r3d_bench_$c.json"))
print("$c value", round(r["value"],1), "exact", round(r.get("exact_basis",{}).get("value",0),1), "surface", round(r.get("autograd_surface",{}).get("value",0),1), "one", round(r.get("one_render_in_flight",{}).get("value",0),1), r["config"]["cameras_per_step"], r["config"]["steps_in_flight"], r["config"]["sh_basis"][150:260])
PY
done
