# One GPU session of the round (run through gpurun from the repo root):  bash tools/gpu_session.sh <tag> [tests|notests] [pmc|nopmc]
# -> gpurun_out/<tag>_*: the -m gpu suite's log, the driver's bench line, a rocprofv3 kernel trace of exactly the timed launches
# (bench.py --only-timed) and the PMC passes of the same command (FETCH_SIZE | WRITE_SIZE | SQ_*, each in its own run).
tag=$1; tests=${2:-tests}; pmc=${3:-pmc}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
if [ "$tests" = tests ]; then
  timeout 1200 python -m pytest tests -m gpu -q --maxfail=12 -p no:cacheprovider > $O/${tag}_gpu_tests.log 2>&1; echo "pytest rc=$?" >> $O/${tag}_gpu_tests.log
  tail -4 $O/${tag}_gpu_tests.log
fi
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/${tag}_bench_cfg2.json 2> $O/${tag}_bench_cfg2.err; echo "bench rc=$?"
python - <<PY
import json
r=json.load(open("$O/${tag}_bench_cfg2.json"))
print("value", round(r["value"],1), "exact", round(r.get("exact_basis",{}).get("value",0),1), "surface", round(r.get("autograd_surface",{}).get("value",0),1),
      "one", round(r.get("one_render_in_flight",{}).get("value",0),1), "bwd_ms", round(r["roofline"]["avg_launch_ms"],3), "alone", round(r["roofline"]["alone_launch_ms"],3),
      "fwd_ms", round(r["roofline"]["fwd_launch_ms"],3), "alone", round(r["roofline"]["alone_fwd_launch_ms"],3))
PY
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}_prof -o prof -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --only-timed > $O/${tag}_bench_cfg2_under_rocprof.json 2> $O/${tag}_prof.err
cd $R
f=$(find $O/${tag}_prof -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/${tag}_bench_cfg2_kernel_stats.csv && head -8 $O/${tag}_bench_cfg2_kernel_stats.csv | cut -c1-200
rm -rf $O/${tag}_prof
if [ "$pmc" = pmc ]; then
  bash tools/pmc.sh ${tag}_fetch "FETCH_SIZE" > /dev/null
  bash tools/pmc.sh ${tag}_write "WRITE_SIZE" > /dev/null
  bash tools/pmc.sh ${tag}_sq "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" | tail -4
  bash tools/pmc.sh ${tag}_sq2 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_BUSY_CYCLES" | tail -4
  rm -rf $O/pmc_${tag}_*/
fi
ls $O | head -40
