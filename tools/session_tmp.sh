cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 900 python -m pytest tests -m gpu -q --maxfail=12 -p no:cacheprovider > $O/r3m_gpu_tests.log 2>&1; echo "pytest rc=$?" >> $O/r3m_gpu_tests.log; tail -3 $O/r3m_gpu_tests.log
L=gsgen_amd/lib_alt
bash tools/ab_libs.sh r3m 2 $L/base.so $L/prev.so - $L/v_asmdpp.so
