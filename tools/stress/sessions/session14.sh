cd $GRAFT_REPO_ROOT
out=gpurun_out/r2n; mkdir -p $out
export TMPDIR=/tmp
echo "== parity subset, default kernels (alive-from-T, masked G)"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_api.py -m gpu -x -q 2>&1 | tail -4
echo "== parity subset, GSGEN_BWD_SH_CHRED=1"
GSGEN_BWD_SH_CHRED=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_api.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -4
for v in 0 1; do
  echo "== bench cfg2 GSGEN_BWD_SH_CHRED=$v"
  GSGEN_BWD_SH_CHRED=$v timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_cfg2_chred$v.json 2> $out/err$v
  python - <<PY
import json
d=json.load(open("$out/bench_cfg2_chred$v.json"))
r=d["roofline"]; o=d["one_render_in_flight"]
print(d["value"], d["ms_per_step"], r["kernel"], "alone bwd", r["alone_launch_ms"], "in-flight bwd", r["avg_launch_ms"], "fwd", r["fwd_launch_ms"], "| one in flight", o["value"], o["bwd_kernel_ms"], o["bwd_kernel"])
PY
done
echo "== cfg3 CHRED=1"
GSGEN_BWD_SH_CHRED=1 timeout 600 python bench.py --config cfg3 --steps 20 --warmup 5 --no-cpu-baseline --no-latency 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
