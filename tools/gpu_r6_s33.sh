# round 6, session 33: the batched sort as one wavefront per tile without LDS (registers up to 1 024 entries, block sort + global merge passes beyond) against the packed four-wavefront workgroups
R=$GRAFT_REPO_ROOT; cd $R; L=$R/gsgen_amd/lib_alt
timeout 600 env GSGEN_HIP_LIB=$L/sort_wave_per_tile.so python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q -p no:cacheprovider -k "sort or binning or long or dense or cfg" 2>&1 | tail -2
bash tools/ab_all.sh r06_s33 2 - $L/sort_wave_per_tile.so
