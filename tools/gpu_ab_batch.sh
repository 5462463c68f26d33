R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -m gpu -q --maxfail=12 -p no:cacheprovider > $O/b_gpu_tests.log 2>&1; echo "pytest rc=$?" >> $O/b_gpu_tests.log
tail -3 $O/b_gpu_tests.log
bash tools/ab.sh "" "--batch 12" "--batch 16" "--batch 4" "" "--batch 12" "--batch 16" "--path heads" "--path heads --batch 12" "--path heads --batch 16" "--path heads" "--path heads --batch 16" "--config cfg3" "--config cfg3 --batch 4" "--config cfg4" "--config cfg4 --batch 16" > /dev/null
cp $O/ab.log $O/b_ab_batch.txt
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-heads --no-surface > $O/b_bench.json 2>/dev/null
python -c "import json;r=json.load(open('$O/b_bench.json'));print('value',round(r['value'],1),'one-render',round(r['one_render_in_flight']['value'],1),'one-step',round(r['one_step_in_flight']['value'],1))"
