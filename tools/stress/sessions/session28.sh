cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
echo "== gpu api tests"
timeout 1500 python -m pytest tests/test_gpu_api.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -4
echo "== host split"
timeout 300 python tools/host_split.py 512 4 2>&1 | tail -1
timeout 300 python tools/host_split.py 256 4 2>&1 | tail -1
echo "== heads path"
for cfg in "--res 512 --batch 4" "--res 512 --batch 8" "--res 800 --batch 8"; do
  echo -n "$cfg: "; timeout 300 python tools/bench_batch.py --no-stats --heads --steps 200 $cfg 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['renders_per_s'],1))"
done
echo "== sh path"
for cfg in "--res 512 --batch 4" "--res 800 --batch 8"; do
  echo -n "$cfg: "; timeout 300 python tools/bench_batch.py --no-stats --steps 200 $cfg 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['renders_per_s'],1))"
done
