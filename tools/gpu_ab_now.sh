R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -m gpu -q --maxfail=12 -p no:cacheprovider > $O/n_gpu_tests.log 2>&1; echo "pytest rc=$?" >> $O/n_gpu_tests.log
tail -3 $O/n_gpu_tests.log
PRE=GSGEN_HIP_LIB=gsgen_amd/lib_alt/pre.so
bash tools/ab.sh "" "$PRE" "" "$PRE" "" "$PRE" "--outlier-fraction 0.01" "$PRE --outlier-fraction 0.01" "--focal-scale 0.7" "$PRE --focal-scale 0.7" "--config cfg4" "$PRE --config cfg4" "--config cfg3" "$PRE --config cfg3" > /dev/null
cp $O/ab.log $O/n_ab.txt
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-heads --no-surface > $O/n_bench.json 2>/dev/null
python -c "import json;r=json.load(open('$O/n_bench.json'));o=r['one_render_in_flight'];print('value',round(r['value'],1),'one-render',round(o['value'],1),o['fwd_kernel_ms'],o['bwd_kernel_ms'],'one-step',round(r['one_step_in_flight']['value'],1))"
