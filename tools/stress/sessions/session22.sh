cd $GRAFT_REPO_ROOT
out=gpurun_out/r2v; mkdir -p $out
export TMPDIR=/tmp
echo "== opt-in matrix-core variants on the GPU"
GSGEN_TEST_MFMA=1 timeout 1200 python -m pytest tests/test_variants.py -m gpu -x -q -k "mfma" 2>&1 | tail -4
echo "== bench cfg2 (walked entries)"
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_cfg2.json 2> $out/bench_err; tail -3 $out/bench_err; python - <<PY
import json
d=json.load(open("$out/bench_cfg2.json")); r=d["roofline"]; o=d["one_render_in_flight"]
print(d["value"], d["ms_per_step"], {k: r.get(k) for k in ("frac","walked_fraction_of_D","walked_bytes_per_launch","walked_achieved_GBs","traffic")}, o["value"], o.get("walked_pairs_per_view"))
PY
timeout 600 python bench.py --config cfg3 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cfg3', d['value'], d['roofline'].get('walked_fraction_of_D'))"
