cd $GRAFT_REPO_ROOT
out=gpurun_out/r2s; mkdir -p $out
export TMPDIR=/tmp
echo "== new tests"
timeout 600 python -m pytest tests/test_gpu_api.py -m gpu -x -q -k "rccl or upload_small" 2>&1 | tail -3
run() {  # $1 = env assignments, $2.. = bench args
  v="$1"; shift
  env $v timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-latency "$@" 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); ro=r['roofline']; print('$v $*', '->', round(r['value'],1), 'r/s  fwd', round(ro['fwd_launch_ms'],3), 'bwd', round(ro['avg_launch_ms'],3), 'alone', round(ro['alone_fwd_launch_ms'],3), round(ro['alone_launch_ms'],3))
"
}
echo "== A/B on cfg2"
run "X=0"
run "GSGEN_PPL_FWD_BATCH=4"
run "GSGEN_PPL_FWD_BATCH=1"
run "X=0" --slots 3
run "X=0" --slots 1
run "X=0" --batch 4 --slots 3
run "X=0" --batch 16
run "GSGEN_BATCH_MAP=0"
echo "== A/B on cfg3"
run "X=0" --config cfg3
run "X=0" --config cfg3 --batch 4 --slots 2
run "X=0" --config cfg3 --batch 2 --slots 4
run "GSGEN_PPL_FWD_BATCH=4" --config cfg3
echo "== A/B on cfg4"
run "X=0" --config cfg4
run "X=0" --config cfg4 --slots 3
run "GSGEN_PPL_FWD_BATCH=4" --config cfg4
