# usage: bash tools/ab_libs.sh <tag> <rounds> <lib or "-" for the in-tree build> ...   -> gpurun_out/<tag>_ab_libs.txt
# same-box A/B of library builds through GSGEN_HIP_LIB: the driver workload with the one-render-in-flight pass, alternating
tag=$1; rounds=$2; shift 2
mkdir -p gpurun_out; out=gpurun_out/${tag}_ab_libs.txt; : > $out
for r in $(seq 1 $rounds); do
  for lib in "$@"; do
    if [ "$lib" = "-" ]; then envs=""; else envs="GSGEN_HIP_LIB=$lib"; fi
    env $envs timeout 240 python bench.py --steps 40 --warmup 10 --no-surface --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); ro=r['roofline']; o=r.get('one_render_in_flight',{})
        print('$lib', 'round $r:', round(r['value'],1),'renders/s  bwd alone',round(ro['alone_launch_ms'],4),'fwd alone',round(ro['alone_fwd_launch_ms'],4),
              'one-render', round(o.get('value',0),1), 'graph', round(o.get('hipgraph_replay',{}).get('value',0),1), 'exact', round(r.get('exact_basis',{}).get('value',0)))
" >> $out
  done
done
cat $out
