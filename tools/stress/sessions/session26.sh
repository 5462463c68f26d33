cd $GRAFT_REPO_ROOT
out=gpurun_out/r2z; mkdir -p $out
export TMPDIR=/tmp
echo "== gpu tests (api + golden + parity)"
timeout 1500 python -m pytest tests/test_gpu_api.py tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_torch_ext.py -m gpu -x -q 2>&1 | tail -4
echo "== heads path"
for cfg in "--res 512 --batch 4" "--res 512 --batch 8" "--res 800 --batch 8"; do
  echo -n "$cfg: "; timeout 300 python tools/bench_batch.py --no-stats --heads --steps 200 $cfg 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['renders_per_s'],1))"
done
