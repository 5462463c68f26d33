#!/bin/bash
# One gpurun call's worth of measurements (run from the repo root on the GPU box):
#   bash tools/stress/gpu_round.sh <tag> [stress_budget_s]
# -> gpurun_out/<tag>/: pytest log, the driver's bench command, A/B bench lines, stress matrices, rocprof stats.
tag=${1:-r2}; budget=${2:-200}
out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
echo "== GPU"; rocm-smi --showproductname 2>/dev/null | grep -i "card series\|gfx" | head -3
echo "== pytest -m gpu"
timeout 1200 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc=$? $(tail -1 $out/pytest.log)"
echo "== bench (driver command)"
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver.json 2> $out/bench_driver.err; echo "rc=$?"
python - <<PY
import json
try:
    d=json.load(open("$out/bench_driver.json"))
    print("value", round(d["value"],1), "ms/step", round(d["ms_per_step"],3), d["timing"], "bwd", d["roofline"]["kernel"], d["roofline"]["avg_launch_ms"], "alone", d["roofline"]["alone_launch_ms"], "fwd", d["roofline"]["fwd_launch_ms"], "one", d.get("one_render_in_flight",{}).get("value"), d.get("one_render_in_flight",{}).get("hipgraph_replay"))
except Exception as e:
    print("bench parse failed", e); print(open("$out/bench_driver.err").read()[-3000:])
PY
