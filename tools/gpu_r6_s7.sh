# round 6, session 7: why is the heads backward 10 % slower inside the model step?  kernel traces of render_heads variants
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
: > $O/r06_s7_heads_variants.txt
for v in "0 0 0" "1 0 0" "0 1 0" "1 1 0" "1 1 1"; do
  tag=$(echo $v | tr -d ' ')
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r06_s7_prof_$tag -o prof -- python $R/tools/prof_heads_variants.py $v > /dev/null 2>> $O/r06_s7_heads_variants.txt
  f=$(find $O/r06_s7_prof_$tag -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && python - "$f" "$v" >> $O/r06_s7_heads_variants.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print("variant", sys.argv[2], " | ".join(f"{r['Name'].split('(')[0].replace('void gs::','').replace('gs::','')[:34]} {float(r['AverageNs'])/1e3:.1f}" for r in rows[:7]))
PY
  rm -rf $O/r06_s7_prof_$tag
done
grep -v amdgpu.ids $O/r06_s7_heads_variants.txt
