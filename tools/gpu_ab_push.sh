# A/B of the push binning of camera batches (tree) against the pull kernels (lib_alt/pre.so = the library of the commit before)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -m gpu -q --maxfail=12 -p no:cacheprovider > $O/p_gpu_tests.log 2>&1; echo "pytest rc=$?" >> $O/p_gpu_tests.log
tail -3 $O/p_gpu_tests.log
PRE=GSGEN_HIP_LIB=gsgen_amd/lib_alt/pre.so
bash tools/ab.sh "" "$PRE" "" "$PRE" "" "$PRE" "--path heads" "$PRE --path heads" "--path heads" "$PRE --path heads" "--path heads" "$PRE --path heads" \
  "--config cfg4" "$PRE --config cfg4" "--config cfg4" "$PRE --config cfg4" "--config cfg3" "$PRE --config cfg3" "--config cfg3" "$PRE --config cfg3" > /dev/null
cp $O/ab.log $O/p_ab_push.txt
