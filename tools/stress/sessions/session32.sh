cd $GRAFT_REPO_ROOT
out=gpurun_out/r2z3; mkdir -p $out
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_api.py -q -x -k "graph_capturable" 2>&1 | tail -15
timeout 200 python tools/bench_step.py > $out/step_eager.json 2> $out/step_err; cat $out/step_eager.json
timeout 200 python tools/bench_step.py --graph > $out/step_graph.json 2>> $out/step_err; cat $out/step_graph.json
timeout 200 python tools/bench_step.py --graph --batch 8 --res 800 > $out/step_graph_8x800.json 2>> $out/step_err; cat $out/step_graph_8x800.json
timeout 200 python tools/bench_step.py --batch 8 --res 800 > $out/step_eager_8x800.json 2>> $out/step_err; cat $out/step_eager_8x800.json
grep -i "error" $out/step_err | head -5
