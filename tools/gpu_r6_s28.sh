# round 6, session 28: long random fuzz on the final tree (sort, per-camera chain, batched SH launches, batched RGB + heads, routed SH) + soak + big scenes
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
export GSGEN_FUZZ_EXAMPLES=${1:-500}
( time timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_sh_bound.py -m gpu -q -p no:cacheprovider -k "fuzz" ) > $O/r06_s28_long_fuzz.log 2>&1; echo "fuzz rc=$?" >> $O/r06_s28_long_fuzz.log; tail -4 $O/r06_s28_long_fuzz.log
timeout 600 python tools/soak.py --iters 300 > $O/r06_s28_soak_sh.txt 2>&1; tail -3 $O/r06_s28_soak_sh.txt
timeout 900 python tools/big_scene.py > $O/r06_s28_big_scene.txt 2>&1; tail -6 $O/r06_s28_big_scene.txt
