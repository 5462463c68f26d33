cd $GRAFT_REPO_ROOT
out=gpurun_out/r2d; mkdir -p $out
echo "== pytest -m gpu (continue past failures)"
timeout 2400 python -m pytest tests -m gpu -q > $out/pytest.log 2>&1; echo "pytest rc=$? $(tail -1 $out/pytest.log)"
grep -n "^FAILED\|^ERROR" $out/pytest.log | head -20
grep -n "AssertionError" $out/pytest.log | head -20
python -m pytest tests/test_gpu_fullsize.py -m gpu -q --durations=5 2>&1 | tail -12
