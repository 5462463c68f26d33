cd $GRAFT_REPO_ROOT
out=gpurun_out/r2q1; mkdir -p $out
export TMPDIR=/tmp
timeout 200 python -m pytest tests/test_gpu_fullsize.py -q -x -k "polynomial" 2>&1 | tail -12
run() {
  env $1 timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-latency $2 > "$out/bench.json" 2> $out/bench.err
  python -c "
import json,sys
d=json.load(open(sys.argv[1])); r=d['roofline']
print(sys.argv[2], sys.argv[3], '|', round(d['value'],1), 'ms/step', round(d['ms_per_step'],3), 'bwd', round(r['avg_launch_ms'],3), 'alone', round(r['alone_launch_ms'],3), 'fwd', round(r['fwd_launch_ms'],3), 'alone fwd', round(r['alone_fwd_launch_ms'],3), r['kernel'][:60])" "$out/bench.json" "$1" "$2" || tail -5 $out/bench.err
}
run "GSGEN_SH_POLY=0" ""
run "GSGEN_SH_POLY=64" ""
run "GSGEN_SH_POLY=64" "--config cfg3"
run "GSGEN_SH_POLY=0" "--config cfg3"
