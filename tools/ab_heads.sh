# usage: bash tools/ab_heads.sh <tag> <rounds> <lib or "-"> ...   -> gpurun_out/<tag>_ab_heads.txt
# same-box A/B of library builds (GSGEN_HIP_LIB) on the RGB + heads path of the driver workload: views/s with three steps in
# flight, with one, and the compositing launches alone / in flight
tag=$1; rounds=$2; shift 2
mkdir -p gpurun_out; out=gpurun_out/${tag}_ab_heads.txt; : > $out
for r in $(seq 1 $rounds); do
  order=("$@"); if [ $((r % 2)) -eq 0 ]; then order=(); for ((i=$#; i>=1; i--)); do order+=("${!i}"); done; fi  # (even rounds in reverse: on some boxes the later run of a pair is the slower one)
  for v in "${order[@]}"; do
    if [ "$v" = "-" ]; then envs=""; else envs="GSGEN_HIP_LIB=$v"; fi
    env $envs timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-surface --no-latency $AB_ARGS 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); h=r.get('heads_path') or {}; ro=h.get('roofline') or {}
        print('$v round $r: sh', round(r['value'],1), 'sh one-step', round((r.get('one_step_in_flight') or {}).get('value',0),1), '| heads', round(h.get('value',0),1), 'one-step', round((h.get('one_step_in_flight') or {}).get('value',0),1),
              'bwd in flight', round(ro.get('avg_launch_ms') or 0,4), 'alone', round(ro.get('alone_launch_ms') or 0,4), 'fwd', round(ro.get('fwd_launch_ms') or 0,4), 'alone', round(ro.get('alone_fwd_launch_ms') or 0,4))
" >> $out
  done
done
cat $out
