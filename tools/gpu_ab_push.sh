# A/B: dynamic LDS for the push binning's counters + push binning per camera (tree) against the commit before (lib_alt/pre.so), and steps in flight
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -m gpu -q --maxfail=12 -p no:cacheprovider > $O/p_gpu_tests.log 2>&1; echo "pytest rc=$?" >> $O/p_gpu_tests.log
tail -3 $O/p_gpu_tests.log
PRE=GSGEN_HIP_LIB=gsgen_amd/lib_alt/pre.so
bash tools/ab.sh "" "$PRE" "--slots 4" "" "$PRE" "--slots 4" "" "$PRE" "--slots 2" "--path heads" "$PRE --path heads" "--path heads --slots 4" "--path heads" "$PRE --path heads" "--path heads --slots 4" \
  "--config cfg4" "$PRE --config cfg4" "--config cfg4 --slots 4" "--config cfg3" "$PRE --config cfg3" > /dev/null
cp $O/ab.log $O/p_ab_push2.txt
