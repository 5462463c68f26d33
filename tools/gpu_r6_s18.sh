# round 6, session 18: the sort's cross-lane steps with ONE comparison per key (no exec-masked branches): model step kernel trace of
# both builds (alone durations: one step at a time), then every line of the driver's command, alternating
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
L=$R/gsgen_amd/lib_alt
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -p no:cacheprovider -k "sort or binning or long_tile" 2>&1 | tail -2
cd /tmp; export TMPDIR=/tmp
for v in sort_512_branchy new; do
  if [ "$v" = new ]; then envs="X=1"; else envs="GSGEN_HIP_LIB=$L/$v.so"; fi
  rm -rf /tmp/prof_$v
  env $envs timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$v -o prof -- python $R/tools/prof_model_step.py 30 > /dev/null 2> $O/r06_s18_model_step_$v.txt
  f=$(find /tmp/prof_$v -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/r06_s18_model_step_kernel_stats_$v.csv
  tail -2 $O/r06_s18_model_step_$v.txt; python - "$O/r06_s18_model_step_kernel_stats_$v.csv" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "gs::" in r["Name"]:
        print(f"   {float(r['AverageNs'])/1e3:9.1f} us x {r['Calls']:>4s}  {r['Name'][:80]}")
PY
done
cd $R
bash tools/ab_all.sh r06_s18 2 $L/sort_512_branchy.so -
