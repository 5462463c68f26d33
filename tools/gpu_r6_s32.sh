# round 6, session 32: workgroup sizes of the small launches IN FLIGHT: batched projection 256 (in tree) / 128 / 64 threads, push binning 512 (in tree) / 256; projection backward at 128 (in tree since session 31)
R=$GRAFT_REPO_ROOT; cd $R; L=$R/gsgen_amd/lib_alt
bash tools/ab_all.sh r06_s32 2 - $L/frame_t128.so $L/frame_t64.so $L/push_t256.so
