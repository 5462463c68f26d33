# round 6, session 25: the trainer-shaped step on the final tree (eager / graph, torch ops vs inside the node), the model step, the geometry chain alone
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
: > $O/r06_s25_bench_step.txt
for r in 1 2; do for v in "--torch-ops" "" "--graph"; do
  echo "round $r ${v:-inside the node}: $(timeout 300 python tools/bench_step.py $v 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(round(j['value'],1),'it/s  ms',round(j['ms_per_iter'],4),'host ms',round(j['host_ms_per_iter'],4),'graph',j['hipgraph'],j.get('hipgraph_error'))")" >> $O/r06_s25_bench_step.txt
done; done
cat $O/r06_s25_bench_step.txt
timeout 300 python tools/prof_model_step.py 30 2> $O/r06_s25_model_step.txt; grep "model step" $O/r06_s25_model_step.txt
cd /tmp; export TMPDIR=/tmp
for w in cfg2 cfg3 cfg4; do
rm -rf /tmp/prof_geo; timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_geo -o prof -- python $R/tools/prof_geometry_chain.py 30 $w > /dev/null 2> $O/r06_s25_geo_$w.txt
f=$(find /tmp/prof_geo -name '*kernel_stats.csv' | head -1); cp $f $O/r06_s25_geometry_chain_${w}_kernel_stats.csv; grep "geometry chain" $O/r06_s25_geo_$w.txt
done
