cd $GRAFT_REPO_ROOT
out=gpurun_out/r2p3; mkdir -p $out
export TMPDIR=/tmp
run() {
  timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-latency "$@" > "$out/bench.json" 2> $out/bench.err
  python -c "
import json,sys
d=json.load(open(sys.argv[1])); r=d['roofline']
print(' '.join(sys.argv[2:]), '|', round(d['value'],1), 'B', d['config']['cameras_per_step'], 'slots', d['config']['steps_in_flight'], 'ms/step', round(d['ms_per_step'],3), 'bwd', round(r['avg_launch_ms'],3), 'alone', round(r['alone_launch_ms'],3), 'fwd', round(r['fwd_launch_ms'],3), round(d['timing']['renders_per_s_min']), round(d['timing']['renders_per_s_max']))" "$out/bench.json" "$@" || tail -5 $out/bench.err
}
for b in 8 12 16 24 32; do run --config cfg2 --batch $b --slots 2; done
run --config cfg2 --batch 16 --slots 3
for b in 16 32 64; do run --config cfg4 --batch $b --slots 2; done
for b in 3 4 6; do run --config cfg3 --batch $b --slots 3; done
