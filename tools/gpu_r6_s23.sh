# round 6, session 23: what the emit's scattered 8-byte stores cost: the geometry chain alone, in-tree against a build whose stores are coalesced (wrong lists: timing only, nothing composited)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
L=$R/gsgen_amd/lib_alt
: > $O/r06_s23_emit_store_cost.txt
for v in new push_dense_stores; do
  if [ "$v" = new ]; then envs="X=1"; else envs="GSGEN_HIP_LIB=$L/$v.so"; fi
  rm -rf /tmp/prof_$v
  env $envs timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$v -o prof -- python $R/tools/prof_geometry_chain.py 30 > /dev/null 2> $O/r06_s23_geo_$v.txt
  f=$(find /tmp/prof_$v -name '*kernel_stats.csv' | head -1)
  echo "$v: $(grep 'geometry chain' $O/r06_s23_geo_$v.txt | tail -1)" >> $O/r06_s23_emit_store_cost.txt; python - "$f" >> $O/r06_s23_emit_store_cost.txt <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "gs::" in r["Name"]:
        print(f"   {float(r['AverageNs'])/1e3:9.1f} us x {r['Calls']:>4s}  {r['Name'][:80]}")
PY
done
cat $O/r06_s23_emit_store_cost.txt
