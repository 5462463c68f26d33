# usage: bash tools/ab.sh "<bench.py args of variant 1>" "<args of variant 2>" ...   -> one line per variant in gpurun_out/ab.log
# (each variant: python bench.py --steps 40 --warmup 10 --only-timed <args>; A/B of kernel shapes: --variant name=value)
mkdir -p gpurun_out; : > gpurun_out/ab.log
for v in "$@"; do
  echo "== $v" >> gpurun_out/ab.log
  timeout 200 python bench.py --steps 40 --warmup 10 --only-timed $v 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print(round(r['value'],1),'renders/s  ms/step',round(r['ms_per_step'],4),'fwd_ms',round(r['roofline']['fwd_launch_ms'],4),'bwd_ms',round(r['roofline']['avg_launch_ms'],4), 'min/max', round(r['timing']['renders_per_s_min']), round(r['timing']['renders_per_s_max']))
" >> gpurun_out/ab.log
done
cat gpurun_out/ab.log
