# A/B of the per-entry exact tier of the polynomial SH kernels (tree: forward held at five wavefronts per SIMD with two scratch
# reloads per entry; lib_alt/nohint.so: the forward at its natural 114 registers, four per SIMD) against the build before it
# (lib_alt/pre.so: any outlier sends its tile to the exact kernel), clean and with 0.1 % / 1 % outlier splats
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -m gpu -q --maxfail=12 -p no:cacheprovider > $O/t_gpu_tests.log 2>&1; echo "pytest rc=$?" >> $O/t_gpu_tests.log
tail -3 $O/t_gpu_tests.log
PRE=GSGEN_HIP_LIB=gsgen_amd/lib_alt/pre.so; NH=GSGEN_HIP_LIB=gsgen_amd/lib_alt/nohint.so
V1=GSGEN_HIP_LIB=gsgen_amd/lib_alt/v1.so
bash tools/ab.sh "" "$V1" "$PRE" "" "$V1" "$PRE" "" "$V1" "$PRE" "--outlier-fraction 0.01" "$V1 --outlier-fraction 0.01" "--config cfg4" "$V1 --config cfg4" > /dev/null
cp $O/ab.log $O/t_ab_tier_v1.txt
