cd $GRAFT_REPO_ROOT
out=gpurun_out/r2l; mkdir -p $out
export TMPDIR=/tmp
echo "== rocprof kernel stats cfg3"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/prof3 -o bench -- python $GRAFT_REPO_ROOT/bench.py --config cfg3 --steps 20 --warmup 5 --no-cpu-baseline --no-latency --slots 1 > $GRAFT_REPO_ROOT/$out/bench_cfg3_rocprof.json 2> $GRAFT_REPO_ROOT/$out/err)
head -14 $out/prof3/bench_kernel_stats.csv | cut -c1-150
echo "== rocprof kernel stats cfg2 one slot (kernel durations without contention)"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/prof2 -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-latency --slots 1 > $GRAFT_REPO_ROOT/$out/bench_cfg2_1slot_rocprof.json 2> $GRAFT_REPO_ROOT/$out/err)
head -14 $out/prof2/bench_kernel_stats.csv | cut -c1-150
echo "== cProfile BatchRenderer 4x512"
timeout 300 python -m cProfile -s tottime tools/bench_batch.py --no-stats --steps 200 --res 512 --batch 4 2>/dev/null | head -45
