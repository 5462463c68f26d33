# round 6, session 9: GPU tests of the SH routing + A/B of the conditional exact-fallback launches (clean scene, 1 % / 5 % outliers, 0.7 x focal)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_sh_bound.py tests/test_gpu_api.py -m gpu -x -q -p no:cacheprovider > $O/r06_s9_gpu_tests_sh.log 2>&1; echo "pytest rc=$?"; tail -3 $O/r06_s9_gpu_tests_sh.log
: > $O/r06_s9_ab_conditional_fallback.txt
for r in 1 2; do
 for stress in "" "--outlier-fraction 0.01" "--outlier-fraction 0.05" "--focal-scale 0.7"; do
  [ $r = 2 ] && [ -n "$stress" ] && continue
  for v in "--always-fallback" ""; do
    timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-surface --no-latency --no-heads --no-other-configs $stress $v 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); ro=r.get('roofline') or {}
        print('${stress:-clean} | ${v:-conditional} round $r: value', round(r['value'],1), 'one-step', round((r.get('one_step_in_flight') or {}).get('value',0),1),
              'bwd in flight', round(ro.get('avg_launch_ms') or 0,4), 'fwd', round(ro.get('fwd_launch_ms') or 0,4), 'frac', round(ro.get('frac') or 0,4), 'tiles exact', r['config'].get('tiles_exact_of_nonempty'), '|', str(r['config'].get('exact_fallback_launches'))[-60:])
" >> $O/r06_s9_ab_conditional_fallback.txt
  done
 done
done
cat $O/r06_s9_ab_conditional_fallback.txt
