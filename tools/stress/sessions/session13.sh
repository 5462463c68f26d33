cd $GRAFT_REPO_ROOT
out=gpurun_out/r2m; mkdir -p $out
export TMPDIR=/tmp
echo "== gpu tests"
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
echo "== BatchRenderer autograd path"
for cfg in "--res 512 --batch 4" "--res 512 --batch 8" "--res 800 --batch 8" "--res 512 --batch 4 --heads" "--res 256 --batch 8"; do
  timeout 300 python tools/bench_batch.py --no-stats --steps 200 $cfg 2>&1 | tail -1
done
echo "== render_frame one camera at a time"
timeout 300 python tools/bench_frame.py 2>&1 | tail -2
echo "== cProfile BatchRenderer 4x512"
timeout 300 python -m cProfile -s tottime tools/bench_batch.py --no-stats --steps 200 --res 512 --batch 4 2>/dev/null | head -30
echo "== bench default"
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_cfg2.json 2> $out/bench_err; tail -c 600 $out/bench_cfg2.json
