# round 6, session 26: gsgen_amd.graph.CapturedStep -- the GPU test, then the trainer-shaped step eager / captured with a fresh camera batch per step
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_api.py -m gpu -x -q -p no:cacheprovider -k "captured_step or raw_parameters or hipgraph or graph" 2>&1 | tail -15
: > $O/r06_s26_bench_step.txt
for r in 1 2; do for v in "" "--graph"; do
  echo "round $r ${v:-eager}: $(timeout 300 python tools/bench_step.py $v 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(round(j['value'],1),'it/s  ms',round(j['ms_per_iter'],4),'host ms',round(j['host_ms_per_iter'],4),'graph',j['hipgraph'],j.get('hipgraph_error'),'captures',j.get('graph_captures'))")" >> $O/r06_s26_bench_step.txt
done; done
cat $O/r06_s26_bench_step.txt
