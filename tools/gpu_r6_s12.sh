# round 6, session 12: raw-parameter mode -- GPU tests, model step, trainer-shape step (eager / graph, torch ops vs inside the node)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_api.py tests/test_gpu_golden.py tests/test_gpu_overflow.py -m gpu -x -q -p no:cacheprovider > $O/r06_s12_gpu_tests.log 2>&1; echo "pytest rc=$?"; tail -3 $O/r06_s12_gpu_tests.log
timeout 300 python tools/prof_model_step.py 30 2> $O/r06_s12_model_step.txt; cat $O/r06_s12_model_step.txt
: > $O/r06_s12_bench_step.txt
for r in 1 2; do for v in "--torch-ops" "" "--torch-ops --graph" "--graph"; do
  echo "round $r ${v:-inside the node}: $(timeout 300 python tools/bench_step.py $v 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(round(j['value'],1),'it/s  ms',round(j['ms_per_iter'],4),'host ms',round(j['host_ms_per_iter'],4),'graph',j['hipgraph'],j.get('hipgraph_error'))")" >> $O/r06_s12_bench_step.txt
done; done
cat $O/r06_s12_bench_step.txt
