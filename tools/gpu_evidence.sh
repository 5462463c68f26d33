# Evidence at one BASELINE config (run through gpurun from the repo root):  bash tools/gpu_evidence.sh <tag> <cfg2|cfg3|cfg4> [full|quick]
# -> gpurun_out/<tag>_bench_<cfg>.json (the whole line incl. cpu_baseline with `full`), <tag>_bench_<cfg>_kernel_stats.csv (rocprofv3
# kernel trace of bench.py --only-timed), pmc_<tag>_<cfg>_{fetch,write,sq}.json (separate --pmc passes of the same command).
tag=$1; cfg=$2; mode=${3:-full}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
extra=""; [ "$mode" = quick ] && extra="--no-cpu-baseline"
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --config $cfg $extra > $O/${tag}_bench_${cfg}.json 2> $O/${tag}_bench_${cfg}.err; echo "bench $cfg rc=$?"
python - <<PY
import json
r=json.load(open("$O/${tag}_bench_${cfg}.json")); h=r.get("heads_path") or {}
print("$cfg value", round(r["value"],1), "exact", round(r.get("exact_basis",{}).get("value",0),1), "one-step", round((r.get("one_step_in_flight") or {}).get("value",0),1),
      "surface", round(r.get("autograd_surface",{}).get("value",0),1), "heads", round(h.get("value",0),1), "cpu", (r.get("cpu_baseline") or {}).get("value"))
PY
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}_prof_$cfg -o prof -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --only-timed --config $cfg > $O/${tag}_bench_${cfg}_under_rocprof.json 2> $O/${tag}_prof_$cfg.err
f=$(find $O/${tag}_prof_$cfg -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/${tag}_bench_${cfg}_kernel_stats.csv && head -6 $O/${tag}_bench_${cfg}_kernel_stats.csv | cut -c1-180
rm -rf $O/${tag}_prof_$cfg
cd $R
export BENCH_ARGS="--config $cfg"
bash tools/pmc.sh ${tag}_${cfg}_fetch "FETCH_SIZE" > /dev/null
bash tools/pmc.sh ${tag}_${cfg}_write "WRITE_SIZE" > /dev/null
bash tools/pmc.sh ${tag}_${cfg}_sq "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" | tail -3
unset BENCH_ARGS
rm -rf $O/pmc_${tag}_${cfg}_*/
