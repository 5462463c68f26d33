cd $GRAFT_REPO_ROOT
out=gpurun_out/r2q2; mkdir -p $out
export TMPDIR=/tmp
for v in "GSGEN_SH_POLY=64" "GSGEN_SH_POLY=0"; do
  env $v timeout 60 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-latency --repeats 3 > "$out/bench_$v.json" 2> $out/bench.err
  python -c "
import json,sys
d=json.load(open(sys.argv[1])); r=d['roofline']
print(sys.argv[2], '|', round(d['value'],1), 'ms/step', round(d['ms_per_step'],3), 'bwd', round(r['avg_launch_ms'],3), 'alone', round(r['alone_launch_ms'],3), 'fwd', round(r['fwd_launch_ms'],3), 'alone fwd', round(r['alone_fwd_launch_ms'],3))" "$out/bench_$v.json" "$v" || tail -3 $out/bench.err
done
