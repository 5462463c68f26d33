cd $GRAFT_REPO_ROOT
out=gpurun_out/r2y; mkdir -p $out
export TMPDIR=/tmp
echo "== gpu tests"
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
echo "== heads path"
for cfg in "--res 512 --batch 4" "--res 512 --batch 8" "--res 800 --batch 8"; do
  echo -n "$cfg: "; timeout 300 python tools/bench_batch.py --no-stats --heads --steps 200 $cfg 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['renders_per_s'],1))"
done
echo "== bench cfg2 sanity"
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-latency 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'])"
