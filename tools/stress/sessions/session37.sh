cd $GRAFT_REPO_ROOT
out=gpurun_out/r2p2; mkdir -p $out
export TMPDIR=/tmp
run() {
  timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-latency "$@" > "$out/bench.json" 2> $out/bench.err
  python -c "
import json,sys
d=json.load(open(sys.argv[1])); r=d['roofline']
print(' '.join(sys.argv[2:]), '|', round(d['value'],1), 'B', d['config']['cameras_per_step'], 'slots', d['config']['steps_in_flight'], 'ms/step', round(d['ms_per_step'],3), 'bwd', round(r['avg_launch_ms'],3), 'alone', round(r['alone_launch_ms'],3), 'fwd', round(r['fwd_launch_ms'],3))" "$out/bench.json" "$@" || tail -5 $out/bench.err
}
for b in 4 6 8 12; do for s in 2 3; do run --config cfg2 --batch $b --slots $s; done; done
for b in 1 2 3 4; do for s in 2 3 4; do run --config cfg3 --batch $b --slots $s; done; done
for b in 4 8 16; do for s in 2 3 4; do run --config cfg4 --batch $b --slots $s; done; done
