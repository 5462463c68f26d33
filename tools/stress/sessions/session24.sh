cd $GRAFT_REPO_ROOT
out=gpurun_out/r2x; mkdir -p $out
export TMPDIR=/tmp
echo "== gpu tests"
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
echo "== heads path A/B"
for v in 1 0; do for cfg in "--res 512 --batch 4" "--res 512 --batch 8" "--res 800 --batch 8"; do
  echo -n "CHAN_PACKED=$v $cfg: "; GSGEN_BWD_CHAN_PACKED=$v timeout 300 python tools/bench_batch.py --no-stats --heads --steps 200 $cfg 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['renders_per_s'],1))"
done; done
echo "== rgb only (C=0) per camera path: bench_rgb"
for v in 1 0; do GSGEN_BWD_CHAN_PACKED=$v timeout 300 python tools/bench_rgb.py 2>&1 | tail -2; done
