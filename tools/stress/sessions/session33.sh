cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( time GSGEN_FUZZ_EXAMPLES=400 timeout 280 python -m pytest tests/test_gpu_parity.py -q -x -k "chain_fuzz" 2>&1 | tail -40 ) 2>&1
