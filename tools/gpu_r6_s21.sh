# round 6, session 21: push binning, one visit at a time (previous build) against four in flight (in tree), both 256 threads: alone durations
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
L=$R/gsgen_amd/lib_alt
for v in push_serial new push_serial new; do
  if [ "$v" = new ]; then envs="X=1"; else envs="GSGEN_HIP_LIB=$L/$v.so"; fi
  rm -rf /tmp/prof_$v
  env $envs timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$v -o prof -- python $R/tools/prof_model_step.py 30 > /dev/null 2> $O/r06_s21_model_step_$v.txt
  f=$(find /tmp/prof_$v -name '*kernel_stats.csv' | head -1)
  echo "$v: $(grep 'model step' $O/r06_s21_model_step_$v.txt | tail -1)"; python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "gs::" in r["Name"] and ("push" in r["Name"] or "sort" in r["Name"]):
        print(f"   {float(r['AverageNs'])/1e3:9.1f} us x {r['Calls']:>4s}  {r['Name'][:80]}")
PY
done
