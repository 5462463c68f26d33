# round 6, session 1: GPU suite on the moment-form tree + same-box A/B of the RGB + heads backward forms
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/r06_s1_gpu_tests.log 2>&1; echo "pytest rc=$?"; tail -5 $O/r06_s1_gpu_tests.log
: > $O/r06_s1_ab_heads_moments.txt
for r in 1 2; do
  for form in plain moments; do
    timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-surface --no-latency --heads-grad-form $form 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); h=r.get('heads_path') or {}; ro=h.get('roofline') or {}
        print('$form round $r: value', round(r['value'],1), 'heads', round(h.get('value',0),1), 'heads one-step', round((h.get('one_step_in_flight') or {}).get('value',0),1),
              'bwd ms in flight', ro.get('avg_launch_ms'), 'alone', ro.get('alone_launch_ms'), 'fwd', ro.get('fwd_launch_ms'), 'alone', ro.get('alone_fwd_launch_ms'))
" >> $O/r06_s1_ab_heads_moments.txt
  done
done
cat $O/r06_s1_ab_heads_moments.txt
