# round 6, session 34: SH batches take their pixel sizes from device memory too (gsgen_sh_view::pixel_size_dev): the headline line of the tree before
# (a copy under _prev_tree: `git archive <rev> | tar -x -C _prev_tree` + that revision's built gsgen_amd/lib/libgsgen_hip.so; the struct grew,
# one Python cannot drive both libraries) against this tree, alternating
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out; out=$O/r06_s34_ab_sh_pixel_size_dev.txt; : > $out
for r in 1 2; do for t in . _prev_tree . _prev_tree; do
  ( cd $R/$t; timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-surface --no-latency --no-other-configs 2>/dev/null ) | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); ro=r['roofline']; h=r.get('heads_path') or {}
        print('[$t] round $r: sh', round(r['value'],1), 'one-step', round(r['one_step_in_flight']['value'],1), 'bwd alone', round(ro['alone_launch_ms'],4), 'fwd alone', round(ro['alone_fwd_launch_ms'],4), '| heads', round(h.get('value',0),1), round((h.get('one_step_in_flight') or {}).get('value',0),1))
" >> $out
done; done
cat $out
