# round 6, session 20: push binning: 256 / 512 / 1024 threads per 2 048-Gaussian chunk (alone durations from the model step's trace, then the A/B)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
L=$R/gsgen_amd/lib_alt
cd /tmp; export TMPDIR=/tmp
for v in new push_t512 push_t1024; do
  if [ "$v" = new ]; then envs="X=1"; else envs="GSGEN_HIP_LIB=$L/$v.so"; fi
  rm -rf /tmp/prof_$v
  env $envs timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$v -o prof -- python $R/tools/prof_model_step.py 30 > /dev/null 2> $O/r06_s20_model_step_$v.txt
  f=$(find /tmp/prof_$v -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/r06_s20_model_step_kernel_stats_$v.csv
  grep "model step" $O/r06_s20_model_step_$v.txt | tail -1; python - "$O/r06_s20_model_step_kernel_stats_$v.csv" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "gs::" in r["Name"] and ("push" in r["Name"] or "scan" in r["Name"]):
        print(f"   {float(r['AverageNs'])/1e3:9.1f} us x {r['Calls']:>4s}  {r['Name'][:80]}")
PY
done
cd $R
GSGEN_HIP_LIB=$L/push_dbg.so python tools/dbg_push_timing.py 2>&1 | tail -4
