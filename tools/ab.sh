# usage: bash tools/ab.sh "ENV1=.. ENV2=.." "ENV..." ...   -> one bench line per variant
mkdir -p gpurun_out; : > gpurun_out/ab.log
for v in "$@"; do
  echo "== $v" >> gpurun_out/ab.log
  env $v timeout 120 python bench.py --steps 64 --warmup 16 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print(round(r['value'],1),'renders/s  fwd_ms',round(r['roofline']['fwd_kernel_ms'],4),'bwd_ms',round(r['roofline']['avg_launch_ms'],4))
" >> gpurun_out/ab.log
done
cat gpurun_out/ab.log
