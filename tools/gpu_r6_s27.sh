# round 6, session 24: a step's backward on a high-priority stream of the slot's own (bench.py --bwd-priority 1) against the default
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
out=$O/r06_s27_ab_tail_priority.txt; : > $out
for r in 1 2; do for v in "" "--tail-priority 1" "--geo-priority 1" "--tail-priority 1 --geo-priority 1"; do
  timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-surface --no-latency --no-other-configs $v 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); ro=r['roofline']; h=r.get('heads_path') or {}; hr=h.get('roofline') or {}
        g=lambda d,k: round((d or {}).get(k) or 0,4)
        print('[$v] round $r: sh', round(r['value'],1), 'one-step', g(r.get('one_step_in_flight'),'value'), 'bwd in flight', g(ro,'avg_launch_ms'), 'frac', g(ro,'frac'), 'fwd', g(ro,'fwd_launch_ms'),
              '| heads', g(h,'value'), 'one-step', g(h.get('one_step_in_flight'),'value'), 'bwd in flight', g(hr,'avg_launch_ms'), 'frac', g(hr,'frac'), 'fwd', g(hr,'fwd_launch_ms'))
" >> $out
done; done
cat $out
