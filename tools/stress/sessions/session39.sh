cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_api.py -q -x -k "32_bits or overflow or capturable" 2>&1 | tail -15
timeout 300 python -m pytest tests/test_gpu_parity.py -q -x 2>&1 | tail -3
