# Round-4 GPU session (run through gpurun from the repo root):  bash tools/gpu_session4.sh <tag> [tests|notests] [pmc|nopmc] [ab|noab]
# -> gpurun_out/<tag>_*: the -m gpu suite's log, the driver's bench line (with heads_path), rocprofv3 kernel traces of exactly the
# timed launches of the SH step and of the RGB + heads step (bench.py --only-timed [--path heads]), PMC passes of the heads step.
tag=$1; tests=${2:-tests}; pmc=${3:-pmc}; ab=${4:-noab}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
if [ "$tests" = tests ]; then
  timeout 1500 python -m pytest tests -m gpu -q --maxfail=12 -p no:cacheprovider > $O/${tag}_gpu_tests.log 2>&1; echo "pytest rc=$?" >> $O/${tag}_gpu_tests.log
  tail -6 $O/${tag}_gpu_tests.log
fi
timeout 500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/${tag}_bench_cfg2.json 2> $O/${tag}_bench_cfg2.err; echo "bench rc=$?"
python - <<PY
import json
r=json.load(open("$O/${tag}_bench_cfg2.json"))
h=r.get("heads_path") or {}
print("value", round(r["value"],1), "exact", round(r.get("exact_basis",{}).get("value",0),1), "surface", round(r.get("autograd_surface",{}).get("value",0),1),
      "one-step", round((r.get("one_step_in_flight") or {}).get("value",0),1),
      "one", round(r.get("one_render_in_flight",{}).get("value",0),1), "bwd_ms", round(r["roofline"]["avg_launch_ms"],3), "alone", round(r["roofline"]["alone_launch_ms"],3),
      "fwd_ms", round(r["roofline"]["fwd_launch_ms"],3), "alone", round(r["roofline"]["alone_fwd_launch_ms"],3))
if h:
    rf=h["roofline"]; print("HEADS value", round(h["value"],1), "one-step", round(h.get("one_step_in_flight",{}).get("value",0),1), "bwd_ms", round(rf["avg_launch_ms"],3), "alone", round(rf.get("alone_launch_ms",0),3),
      "fwd_ms", round(rf["fwd_launch_ms"],3), "alone", round(rf.get("alone_fwd_launch_ms",0),3), "frac", round(rf["frac"],3), "host_ms", round(h["host_enqueue_ms_per_step"],3))
PY
cd /tmp && export TMPDIR=/tmp
for path in sh heads; do
  sfx=""; [ $path = heads ] && sfx="_heads"
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}_prof$sfx -o prof -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --only-timed --path $path > $O/${tag}_bench_cfg2${sfx}_under_rocprof.json 2> $O/${tag}_prof$sfx.err
  f=$(find $O/${tag}_prof$sfx -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/${tag}_bench_cfg2${sfx}_kernel_stats.csv && head -9 $O/${tag}_bench_cfg2${sfx}_kernel_stats.csv | cut -c1-200
  rm -rf $O/${tag}_prof$sfx
done
cd $R
if [ "$pmc" = pmc ]; then
  export BENCH_ARGS="--path heads"
  bash tools/pmc.sh ${tag}_heads_fetch "FETCH_SIZE" > /dev/null
  bash tools/pmc.sh ${tag}_heads_write "WRITE_SIZE" > /dev/null
  bash tools/pmc.sh ${tag}_heads_sq "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" | tail -4
  bash tools/pmc.sh ${tag}_heads_sq2 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_BUSY_CYCLES" | tail -4
  unset BENCH_ARGS
  rm -rf $O/pmc_${tag}_*/
fi
if [ "$ab" = ab ]; then
  bash tools/ab.sh "" "--torch-fill" "" "--torch-fill" "--path heads" "--path heads --torch-fill" > /dev/null; cp $O/ab.log $O/${tag}_ab_fill.txt; cat $O/${tag}_ab_fill.txt
fi
ls $O | head -60
