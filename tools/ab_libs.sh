# usage: bash tools/ab_libs.sh <tag> <rounds> <variant> ...   -> gpurun_out/<tag>_ab_libs.txt
# variant = <lib or "-" for the in-tree build>[@<extra bench.py args, '+' for spaces>]
# same-box A/B of library builds through GSGEN_HIP_LIB: the driver workload with the one-render-in-flight pass, alternating
tag=$1; rounds=$2; shift 2
mkdir -p gpurun_out; out=gpurun_out/${tag}_ab_libs.txt; : > $out
for r in $(seq 1 $rounds); do
  for v in "$@"; do
    lib=${v%%@*}; extra=""; [ "$lib" != "$v" ] && extra=$(echo "${v#*@}" | tr '+' ' ')
    if [ "$lib" = "-" ]; then envs=""; else envs="GSGEN_HIP_LIB=$lib"; fi
    env $envs timeout 240 python bench.py --steps 40 --warmup 10 --no-surface --no-cpu-baseline $extra 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); ro=r['roofline']; o=r.get('one_render_in_flight') or {}
        print('$v', 'round $r:', round(r['value'],1),'renders/s  bwd alone',round(ro['alone_launch_ms'],4),'fwd alone',round(ro['alone_fwd_launch_ms'],4),
              'one-step', round((r.get('one_step_in_flight') or {}).get('value',0),1), 'heads', round((r.get('heads_path') or {}).get('value',0),1), 'one-render', round(o.get('value',0),1), 'graph', round((o.get('hipgraph_replay') or {}).get('value',0),1), 'exact', round((r.get('exact_basis') or {}).get('value',0)))
" >> $out
  done
done
cat $out
