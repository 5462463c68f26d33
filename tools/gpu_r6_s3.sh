# round 6, session 3: GPU suite on the SH moment-form tree + same-box A/B of the SH backward forms (and the heads path after session 2's picks)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/r06_s3_gpu_tests.log 2>&1; echo "pytest rc=$?"; tail -3 $O/r06_s3_gpu_tests.log
export AB_ARGS=""
: > $O/r06_s3_ab_sh_moments.txt
for r in 1 2; do
  for form in plain moments; do
    timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-surface --no-latency --sh-grad-form $form 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); h=r.get('heads_path') or {}; ro=r.get('roofline') or {}
        print('sh $form round $r: value', round(r['value'],1), 'one-step', round((r.get('one_step_in_flight') or {}).get('value',0),1), 'exact', round((r.get('exact_basis') or {}).get('value',0),1),
              'bwd ms in flight', round(ro.get('avg_launch_ms') or 0,4), 'alone', round(ro.get('alone_launch_ms') or 0,4), 'fwd', round(ro.get('fwd_launch_ms') or 0,4), 'alone', round(ro.get('alone_fwd_launch_ms') or 0,4), 'frac', ro.get('frac'),
              '| heads', round(h.get('value',0),1), 'one-step', round((h.get('one_step_in_flight') or {}).get('value',0),1))
" >> $O/r06_s3_ab_sh_moments.txt
  done
done
cat $O/r06_s3_ab_sh_moments.txt
