# usage: bash tools/ab.sh "[ENV=val ...] [-- ]<bench.py args of variant 1>" "<variant 2>" ...   -> one line per variant in gpurun_out/ab.log
# each variant: [env ...] python bench.py --steps 40 --warmup 10 --no-latency --no-surface --no-cpu-baseline <args>
# (A/B of kernel shapes: --variant name=value; of experiment builds of the library: GSGEN_HIP_LIB=gsgen_amd/lib_alt/<build>.so)
mkdir -p gpurun_out; : > gpurun_out/ab.log
for v in "$@"; do
  echo "== $v" >> gpurun_out/ab.log
  envs=""; args=""
  for w in $v; do case "$w" in *=*) if [ -z "$args" ] && [[ "$w" != --* ]]; then envs="$envs $w"; else args="$args $w"; fi;; --) ;; *) args="$args $w";; esac; done
  env $envs timeout 200 python bench.py --steps 40 --warmup 10 --no-latency --no-surface --no-cpu-baseline --no-heads $args 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); ro=r['roofline']; print(round(r['value'],1),'renders/s  ms/step',round(r['ms_per_step'],4),'fwd_ms',round(ro['fwd_launch_ms'],4),'bwd_ms',round(ro['avg_launch_ms'],4),'alone fwd',round(ro['alone_fwd_launch_ms'],4),'bwd',round(ro['alone_launch_ms'],4),'min/max', round(r['timing']['renders_per_s_min']), round(r['timing']['renders_per_s_max']), 'exact', round(r.get('exact_basis',{}).get('value',0)))
" >> gpurun_out/ab.log
done
cat gpurun_out/ab.log
