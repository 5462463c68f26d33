cd $GRAFT_REPO_ROOT
out=gpurun_out/r2r; mkdir -p $out
export TMPDIR=/tmp
echo "== bench cfg2, forced single-rank process group (gather on the communication stream)"
GSGEN_BENCH_FORCE_DIST=1 timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-latency > $out/bench_cfg2_forcedist.json 2> $out/err_fd; tail -3 $out/err_fd
python - <<PY
import json
d=json.load(open("$out/bench_cfg2_forcedist.json")); print(d["value"], d["ms_per_step"], d["config"]["gather"], d["timing"]["host_enqueue_us_per_step_by_call"])
PY
echo "== same under torchrun with 1 rank"
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-latency 2> $out/err_tr | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['n_gpus'], d['config']['gather'])"
tail -2 $out/err_tr
echo "== bench cfg2 plain"
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_cfg2.json 2> $out/bench_err; python - <<PY
import json
d=json.load(open("$out/bench_cfg2.json")); r=d["roofline"]; o=d["one_render_in_flight"]
print(d["value"], d["ms_per_step"], "alone", r["alone_launch_ms"], "traffic", r["traffic"], "valu", r.get("alone_valu_frac"), "| one in flight", o["value"])
PY
