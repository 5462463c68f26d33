# A/B of the per-entry exact tier of the polynomial SH kernels (tree: forward held at five wavefronts per SIMD with two scratch
# reloads per entry; lib_alt/nohint.so: the forward at its natural 114 registers, four per SIMD) against the build before it
# (lib_alt/pre.so: any outlier sends its tile to the exact kernel), clean and with 0.1 % / 1 % outlier splats
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests/test_gpu_sh_bound.py tests/test_gpu_fullsize.py -m gpu -q --maxfail=12 -p no:cacheprovider > $O/t_gpu_tests.log 2>&1; echo "pytest rc=$?" >> $O/t_gpu_tests.log
tail -3 $O/t_gpu_tests.log
PRE=GSGEN_HIP_LIB=gsgen_amd/lib_alt/pre.so; NH=GSGEN_HIP_LIB=gsgen_amd/lib_alt/nohint.so
bash tools/ab.sh "" "$PRE" "" "$PRE" "" "$PRE" \
  "--outlier-fraction 0.001" "$PRE --outlier-fraction 0.001" \
  "--outlier-fraction 0.01" "$PRE --outlier-fraction 0.01" "--outlier-fraction 0.05" \
  "--config cfg4" "$PRE --config cfg4" "--config cfg4" "$PRE --config cfg4" > /dev/null
cp $O/ab.log $O/t_ab_tier.txt
