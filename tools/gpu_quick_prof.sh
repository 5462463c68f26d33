# quick look: GPU suite, then kernel traces of the SH step and the RGB + heads step (three steps in flight and one)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -m gpu -q --maxfail=12 -p no:cacheprovider > $O/q_gpu_tests.log 2>&1; echo "pytest rc=$?" >> $O/q_gpu_tests.log
tail -3 $O/q_gpu_tests.log
cd /tmp && export TMPDIR=/tmp
for v in "sh 0" "heads 0" "sh 1" "heads 1"; do set -- $v
  sl=""; [ $2 = 1 ] && sl="--slots 1"
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/q_prof -o prof -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --only-timed --path $1 $sl > $O/q_$1_$2.json 2> $O/q_prof.err
  f=$(find $O/q_prof -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/q_$1_slots$2_kernel_stats.csv
  rm -rf $O/q_prof
done
