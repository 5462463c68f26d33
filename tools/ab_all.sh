# usage: bash tools/ab_all.sh <tag> <rounds> <lib or "-"> ...   -> gpurun_out/<tag>_ab_all.txt
# same-box A/B of library builds (GSGEN_HIP_LIB) through the driver's own command minus the CPU baseline: the SH value, the RGB + heads
# path, the model-level call and the other configurations (cfg3 / cfg4 / stress lines)
tag=$1; rounds=$2; shift 2
mkdir -p gpurun_out; out=gpurun_out/${tag}_ab_all.txt; : > $out
for r in $(seq 1 $rounds); do
  order=("$@"); if [ $((r % 2)) -eq 0 ]; then order=(); for ((i=$#; i>=1; i--)); do order+=("${!i}"); done; fi  # (even rounds in reverse: on some boxes the later run of a pair is the slower one)
  for v in "${order[@]}"; do
    if [ "$v" = "-" ]; then envs="X=1"; else envs="GSGEN_HIP_LIB=$v"; fi
    env $envs timeout 600 python bench.py --no-cpu-baseline --no-latency $AB_ARGS 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); h=r.get('heads_path') or {}; oc=r.get('other_configs') or {}; ms=r.get('model_surface') or {}
        g=lambda d,k: round((d or {}).get(k) or 0,1)
        print('$(basename $v) round $r: sh', round(r['value'],1), 'one-step', g(r.get('one_step_in_flight'),'value'), '| heads', g(h,'value'), 'one-step', g(h.get('one_step_in_flight'),'value'),
              '| model', g(ms,'value'), 'drop-in', g(r.get('dropin_gs_surface'),'value'), '|', ' '.join(f'{k} {g(v,\"value\")}' for k,v in oc.items()))
" >> $out
  done
done
cat $out
