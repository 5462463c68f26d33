# round 6, session 17: the whole GPU suite on the tree with the packed sort launch (one wavefront per list below 512 entries)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/r06_s17_gpu_tests.log 2>&1; echo "pytest rc=$?"; tail -3 $O/r06_s17_gpu_tests.log
