# usage: bash tools/ab_sh.sh <tag> <rounds> <lib or "-"> ...   -> gpurun_out/<tag>_ab_sh.txt : same-box A/B of library builds on the SH path of the driver workload
tag=$1; rounds=$2; shift 2
mkdir -p gpurun_out; out=gpurun_out/${tag}_ab_sh.txt; : > $out
for r in $(seq 1 $rounds); do
  order=("$@"); if [ $((r % 2)) -eq 0 ]; then order=(); for ((i=$#; i>=1; i--)); do order+=("${!i}"); done; fi  # (even rounds in reverse: on some boxes the later run of a pair is the slower one)
  for v in "${order[@]}"; do
    if [ "$v" = "-" ]; then envs=""; else envs="GSGEN_HIP_LIB=$v"; fi
    env $envs timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-surface --no-latency --no-heads --no-other-configs $AB_ARGS 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); ro=r.get('roofline') or {}
        print('$v round $r: sh', round(r['value'],1), 'one-step', round((r.get('one_step_in_flight') or {}).get('value',0),1), 'exact', round((r.get('exact_basis') or {}).get('value',0),1),
              'bwd in flight', round(ro.get('avg_launch_ms') or 0,4), 'alone', round(ro.get('alone_launch_ms') or 0,4), 'fwd', round(ro.get('fwd_launch_ms') or 0,4), 'alone', round(ro.get('alone_fwd_launch_ms') or 0,4), 'frac', round(ro.get('frac') or 0,4))
" >> $out
  done
done
cat $out
