cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 900 python -m pytest tests -m gpu -q --maxfail=12 -p no:cacheprovider > $O/r3o_gpu_tests.log 2>&1; echo "pytest rc=$?" >> $O/r3o_gpu_tests.log; tail -3 $O/r3o_gpu_tests.log
bash tools/ab_libs.sh r3o 1 - -@--latency-segments+4 -@--latency-segments+16 -@--variant+ppl_fwd_poly=4 -@--slots+2 -@--slots+4
for c in 1 3 4; do timeout 300 python bench.py --config cfg$c --steps 20 --warmup 5 --no-cpu-baseline > $O/r3o_bench_cfg$c.json 2> $O/r3o_bench_cfg$c.err; python -c "
import json; r=json.load(open('$O/r3o_bench_cfg$c.json')); print('cfg$c', round(r['value'],1), 'exact', round((r.get('exact_basis') or {}).get('value',0),1), 'surface', round((r.get('autograd_surface') or {}).get('value',0),1), 'one', round((r.get('one_render_in_flight') or {}).get('value',0),1))"; done
