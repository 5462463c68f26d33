# round 6, session 16: where the host time of the trainer-shaped eager step goes (cProfile + torch profiler), threshold 512 in-tree
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
python tools/bench_step.py --cprofile > $O/r06_s16_cprofile.json 2> $O/r06_s16_cprofile.txt; cat $O/r06_s16_cprofile.json
python tools/bench_step.py --profile > $O/r06_s16_torchprof.json 2> $O/r06_s16_torchprof.txt; cat $O/r06_s16_torchprof.json
python tools/bench_step.py; python tools/bench_step.py --graph
