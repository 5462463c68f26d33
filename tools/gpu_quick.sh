R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -m gpu -q --maxfail=12 -p no:cacheprovider > $O/q_gpu_tests.log 2>&1; echo "pytest rc=$?" >> $O/q_gpu_tests.log
tail -3 $O/q_gpu_tests.log
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-heads --no-surface > $O/q_bench.json 2>/dev/null
python -c "import json;r=json.load(open('$O/q_bench.json'));o=r['one_render_in_flight'];print('value',round(r['value'],1),'one-render',round(o['value'],1),o['fwd_kernel_ms'],o['bwd_kernel_ms'],'one-step',round(r['one_step_in_flight']['value'],1))"
python -c "from __graft_entry__ import smoke; smoke(); print('smoke ok')"
