# round 6, session 35: the one-wavefront-per-tile sort again, four rounds with the order alternating (session 33 ran the in-tree build first in every pair)
R=$GRAFT_REPO_ROOT; cd $R; L=$R/gsgen_amd/lib_alt
bash tools/ab_all.sh r06_s35 4 - $L/sort_wave_per_tile.so
