# Final evidence of a round in one gpurun call:  bash tools/gpu_final.sh <tag>
# driver bench line + traces + heads PMC (gpu_session4), per-config bench lines + traces + PMC (gpu_evidence), routing stress lines
tag=$1; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
bash tools/gpu_session4.sh ${tag}f notests pmc noab
for cfg in cfg2 cfg3 cfg4; do bash tools/gpu_evidence.sh ${tag}e $cfg full; done
cd $R
for st in "focal_0.7 --focal-scale 0.7" "outliers_0.001 --outlier-fraction 0.001" "outliers_0.01 --outlier-fraction 0.01" "outliers_0.05 --outlier-fraction 0.05"; do set -- $st
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-latency --no-surface --no-heads $2 $3 > $O/${tag}_bench_cfg2_stress_$1.json 2>/dev/null
  python -c "import json;r=json.load(open('$O/${tag}_bench_cfg2_stress_$1.json'));print('$1',round(r['value'],1),'exact',round(r['exact_basis']['value'],1),r['config']['tiles_exact_of_nonempty'])"
done
ls $O | wc -l
