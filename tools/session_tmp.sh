cd $GRAFT_REPO_ROOT
for lib in "" gsgen_amd/lib_alt/sort_coop_batch.so; do for c in cfg3 cfg4; do GSGEN_HIP_LIB=$lib timeout 300 python bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline --no-surface --no-latency 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print('$c [$lib]', round(r['value'],1), 'exact', round((r.get('exact_basis') or {}).get('value',0),1), 'maxlen', r['config']['list_length_histogram']['max_length'])"; done; done
cd /tmp && export TMPDIR=/tmp
GSGEN_HIP_LIB=$GRAFT_REPO_ROOT/gsgen_amd/lib_alt/sort_coop_batch.so timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3z_prof -o prof -- python $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --only-timed > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/r3z_prof -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && grep "sort_tiles" $f | cut -c1-60,100-260
rm -rf gpurun_out/r3z_prof
