# round 6, session 22: push binning threads per workgroup (256 in tree / 512 / 1024) and 1 024-Gaussian chunks: alone durations, then every line
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
L=$R/gsgen_amd/lib_alt
: > $O/r06_s22_push_alone.txt
for v in new push_t512 push_t1024 push_c1024_t512; do
  if [ "$v" = new ]; then envs="X=1"; else envs="GSGEN_HIP_LIB=$L/$v.so"; fi
  rm -rf /tmp/prof_$v
  env $envs timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$v -o prof -- python $R/tools/prof_model_step.py 30 > /dev/null 2> $O/r06_s22_model_step_$v.txt
  f=$(find /tmp/prof_$v -name '*kernel_stats.csv' | head -1)
  echo "$v: $(grep 'model step' $O/r06_s22_model_step_$v.txt | tail -1)" >> $O/r06_s22_push_alone.txt; python - "$f" >> $O/r06_s22_push_alone.txt <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "gs::" in r["Name"] and ("push" in r["Name"] or "scan" in r["Name"]):
        print(f"   {float(r['AverageNs'])/1e3:9.1f} us x {r['Calls']:>4s}  {r['Name'][:80]}")
PY
done
cat $O/r06_s22_push_alone.txt
cd $R
bash tools/ab_all.sh r06_s22 2 - $L/push_t512.so $L/push_t1024.so $L/push_c1024_t512.so
