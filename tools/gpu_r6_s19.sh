# round 6, session 19: push binning with four visits in flight per lane: binning tests, model step kernel trace (alone durations), A/B
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
L=$R/gsgen_amd/lib_alt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_overflow.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -2
cd /tmp; export TMPDIR=/tmp
for v in push_serial new; do
  if [ "$v" = new ]; then envs="X=1"; else envs="GSGEN_HIP_LIB=$L/$v.so"; fi
  rm -rf /tmp/prof_$v
  env $envs timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$v -o prof -- python $R/tools/prof_model_step.py 30 > /dev/null 2> $O/r06_s19_model_step_$v.txt
  f=$(find /tmp/prof_$v -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/r06_s19_model_step_kernel_stats_$v.csv
  grep "ms per step\|views/s" $O/r06_s19_model_step_$v.txt | tail -2; python - "$O/r06_s19_model_step_kernel_stats_$v.csv" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "gs::" in r["Name"] and float(r["AverageNs"]) > 6000:
        print(f"   {float(r['AverageNs'])/1e3:9.1f} us x {r['Calls']:>4s}  {r['Name'][:80]}")
PY
done
cd $R
bash tools/ab_all.sh r06_s19 2 $L/push_serial.so -
