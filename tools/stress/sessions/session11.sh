cd $GRAFT_REPO_ROOT
out=gpurun_out/r2k; mkdir -p $out
export TMPDIR=/tmp
echo "== quick GPU tests of the touched paths"
timeout 900 python -m pytest tests/test_gpu_api.py -m gpu -q -x 2>&1 | tail -2
echo "== BatchRenderer (public autograd path)"
for a in "--res 512 --batch 4" "--res 512 --batch 8" "--res 800 --batch 8" "--res 512 --batch 4 --heads"; do
  timeout 300 python tools/bench_batch.py --no-stats --steps 30 $a 2>/dev/null | tail -1
done
echo "== bench, all configs"
for c in cfg2 cfg3 cfg4 cfg1; do
  timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --config $c > $out/bench_$c.json 2> $out/bench_$c.err
  python -c "
import json
d=json.load(open('$out/bench_$c.json')); r=d['roofline']; o=d.get('one_render_in_flight',{})
print('$c', round(d['value'],1), 'B', d['config']['cameras_per_step'], 'x', d['config']['steps_in_flight'], 'bwd', round(r['avg_launch_ms'],3), 'alone', round(r['alone_launch_ms'],3), 'fwd', round(r['fwd_launch_ms'],3), 'alone fwd', round(r['alone_fwd_launch_ms'],3), '| one', round(o.get('value',0),1), 'graph', round(o.get('hipgraph_replay',{}).get('value',0),1), '| cpu', round(d['cpu_baseline']['value'],3))" || tail -5 $out/bench_$c.err
done
echo "== rocprof kernel stats of the driver's bench command"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-latency > $GRAFT_REPO_ROOT/$out/bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/$out/bench_under_rocprof.err)
head -8 $out/prof/bench_kernel_stats.csv | cut -c1-170
echo "== PMC passes"
bash tools/pmc.sh r2k_fetch "FETCH_SIZE" | cut -c1-200
bash tools/pmc.sh r2k_write "WRITE_SIZE" | cut -c1-200
bash tools/pmc.sh r2k_sq "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" | cut -c1-300
echo "== stress matrix (third box)"
python tools/stress/run_matrix.py --out $out/stress_default.jsonl --procs 12 --budget-s 120 --launches 4
LD_LIBRARY_PATH=gsgen_amd/lib_alt/plain python tools/stress/run_matrix.py --out $out/stress_plain.jsonl --procs 60 --budget-s 90 --variants mfma2,mfma4 --batches 0,3 --Cs 1,4 --orders test-first --launches 3 --extra "--dump-bad 16"
