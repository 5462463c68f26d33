# A/B of the round-4 change "view parameter blocks in the kernel arguments" (tree) against the build before it (lib_alt/old.so:
# k_write_params + table in device memory) and against the RGB + heads forward at five wavefronts per SIMD (lib_alt/f5.so)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -m gpu -q --maxfail=12 -p no:cacheprovider > $O/k_gpu_tests.log 2>&1; echo "pytest rc=$?" >> $O/k_gpu_tests.log
tail -3 $O/k_gpu_tests.log
OLD=GSGEN_HIP_LIB=gsgen_amd/lib_alt/old.so; F5=GSGEN_HIP_LIB=gsgen_amd/lib_alt/f5.so
bash tools/ab.sh "" "$OLD" "" "$OLD" "" "$OLD" \
  "--path heads" "$OLD --path heads" "$F5 --path heads" "--path heads" "$OLD --path heads" "$F5 --path heads" "--path heads" "$OLD --path heads" "$F5 --path heads" \
  "--config cfg4" "$OLD --config cfg4" "--config cfg4" "$OLD --config cfg4" > /dev/null
cp $O/ab.log $O/k_ab_kernarg.txt
