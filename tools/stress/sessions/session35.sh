cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( time GSGEN_FUZZ_EXAMPLES=${1:-25} timeout 280 python -m pytest tests/test_gpu_fullsize.py -q -x -k "heads_fuzz" 2>&1 | tail -40 ) 2>&1
