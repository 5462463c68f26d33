# round 6, session 6: prepared evaluation records (chol) A/B on the heads path + GPU suite + model step
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/r06_s6_gpu_tests.log 2>&1; echo "pytest rc=$?"; tail -3 $O/r06_s6_gpu_tests.log
: > $O/r06_s6_ab_heads_chol.txt
for r in 1 2; do
  for v in "--no-heads-chol" ""; do
    timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-surface --no-latency $v 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); h=r.get('heads_path') or {}; ro=h.get('roofline') or {}
        print('chol ${v:-on} round $r: sh', round(r['value'],1), '| heads', round(h.get('value',0),1), 'one-step', round((h.get('one_step_in_flight') or {}).get('value',0),1),
              'bwd in flight', round(ro.get('avg_launch_ms') or 0,4), 'alone', round(ro.get('alone_launch_ms') or 0,4), 'fwd', round(ro.get('fwd_launch_ms') or 0,4), 'alone', round(ro.get('alone_fwd_launch_ms') or 0,4))
" >> $O/r06_s6_ab_heads_chol.txt
  done
done
cat $O/r06_s6_ab_heads_chol.txt
timeout 300 python tools/prof_model_step.py 30 2> $O/r06_s6_model_step.txt; cat $O/r06_s6_model_step.txt
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r06_s6_prof -o prof -- python $R/tools/prof_model_step.py 30 > /dev/null 2> $O/r06_s6_prof.err
f=$(find $O/r06_s6_prof -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/r06_s6_model_step_kernel_stats.csv && head -12 $O/r06_s6_model_step_kernel_stats.csv | cut -c1-150
rm -rf $O/r06_s6_prof
