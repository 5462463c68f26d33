# round 6, session 31: k_project_bwd_views with 64 / 128 / 256 threads per 64 Gaussians (1 / 2 / 4 view lanes), every line of the driver's command
R=$GRAFT_REPO_ROOT; cd $R; L=$R/gsgen_amd/lib_alt
bash tools/ab_all.sh r06_s31 2 - $L/pbv_t128.so $L/pbv_t64.so
