mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -q 2>&1 | tail -8 > gpurun_out/pytest_gpu.log
for f in 4 2 1; do for b in 4 2 1; do
  echo "PPL fwd=$f bwd=$b" >> gpurun_out/sweep.log
  timeout 120 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --variant ppl_fwd=$f --variant ppl_bwd=$b 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print(round(r['value'],1),'renders/s  fwd_ms',round(r['roofline']['fwd_kernel_ms'],4),'bwd_ms',round(r['roofline']['avg_launch_ms'],4))
" >> gpurun_out/sweep.log
done; done
tail -8 gpurun_out/pytest_gpu.log; cat gpurun_out/sweep.log
