# round 6, session 4: where the model-level step's time goes (kernel trace of tools/prof_model_step.py)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 300 python tools/prof_model_step.py 30 2> $O/r06_s4_model_step.txt; cat $O/r06_s4_model_step.txt
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r06_s4_prof -o prof -- python $R/tools/prof_model_step.py 30 > /dev/null 2> $O/r06_s4_prof.err
f=$(find $O/r06_s4_prof -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/r06_s4_model_step_kernel_stats.csv && head -45 $O/r06_s4_model_step_kernel_stats.csv | cut -c1-200
rm -rf $O/r06_s4_prof
