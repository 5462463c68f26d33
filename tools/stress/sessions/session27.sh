cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
echo "== cProfile heads 4x256"
timeout 300 python -m cProfile -s tottime tools/bench_batch.py --no-stats --heads --steps 300 --res 256 --batch 4 2>/dev/null | head -40
echo "== cProfile sh 4x256"
timeout 300 python -m cProfile -s tottime tools/bench_batch.py --no-stats --steps 300 --res 256 --batch 4 2>/dev/null | head -30
