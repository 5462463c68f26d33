# round 6, session 15: the sort launch's one-wavefront threshold: previous build / 256 (in-tree) / 512 / 1024, every line of the driver's command
R=$GRAFT_REPO_ROOT; cd $R
L=$R/gsgen_amd/lib_alt
bash tools/ab_all.sh r06_s15 2 $L/sort_prev.so $L/sort_512.so $L/sort_1024.so
