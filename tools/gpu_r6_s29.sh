# round 6, session 29: steps in flight (2 / 3 / 4) on the final tree
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
out=$O/r06_s29_ab_slots.txt; : > $out
for r in 1 2; do for v in "--slots 2" "--slots 3" "--slots 4" "--slots 5"; do
  timeout 400 python bench.py --steps 24 --warmup 6 --no-cpu-baseline --no-surface --no-latency --no-other-configs $v 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); ro=r['roofline']; h=r.get('heads_path') or {}; hr=h.get('roofline') or {}
        g=lambda d,k: round((d or {}).get(k) or 0,4)
        print('[$v] round $r: sh', round(r['value'],1), 'bwd in flight', g(ro,'avg_launch_ms'), 'frac', g(ro,'frac'), '| heads', g(h,'value'), 'bwd in flight', g(hr,'avg_launch_ms'))
" >> $out
done; done
cat $out
