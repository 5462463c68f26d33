# round 6, session 8: the driver's command as the driver runs it (wall clock), heads variants again after the background fix
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
t0=$(date +%s.%N)
timeout 900 python bench.py > $O/r06_s8_bench_cfg2.json 2> $O/r06_s8_bench_cfg2.err; echo "bench rc=$? wall $(python -c "import time;print(round(time.time()-$t0,1))") s"
python - <<PY
import json
r=json.load(open("$O/r06_s8_bench_cfg2.json")); h=r.get("heads_path") or {}
print("value", round(r["value"],1), "ms/step", round(r["ms_per_step"],4), "frac", r["roofline"].get("frac"), "whole", r["roofline"].get("whole_render_hbm_frac"), "one-step", round((r.get("one_step_in_flight") or {}).get("value",0),1), "exact", round((r.get("exact_basis") or {}).get("value",0),1))
print("heads", round(h.get("value",0),1), "one-step", round((h.get("one_step_in_flight") or {}).get("value",0),1), "frac", (h.get("roofline") or {}).get("frac"))
print("autograd", round((r.get("autograd_surface") or {}).get("value",0),1), "model", round((r.get("model_surface") or {}).get("value",0),1), "dropin", round((r.get("dropin_gs_surface") or {}).get("value",0),1), "cpu", (r.get("cpu_baseline") or {}).get("value"))
for k,v in (r.get("other_configs") or {}).items(): print(k, {a: (round(b,3) if isinstance(b,float) else b) for a,b in v.items() if a in ("value","ms_per_step","roofline_frac","whole_render_hbm_frac","wall_s","error","cameras_per_step")})
PY
cd /tmp && export TMPDIR=/tmp
: > $O/r06_s8_heads_variants.txt
for v in "0 0 0" "1 1 1"; do
  tag=$(echo $v | tr -d ' ')
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r06_s8_prof_$tag -o prof -- python $R/tools/prof_heads_variants.py $v > /dev/null 2>> $O/r06_s8_heads_variants.txt
  f=$(find $O/r06_s8_prof_$tag -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && python - "$f" "$v" >> $O/r06_s8_heads_variants.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print("variant", sys.argv[2], " | ".join(f"{r['Name'].split('(')[0].replace('void gs::','').replace('gs::','')[:34]} {float(r['AverageNs'])/1e3:.1f}" for r in rows[:7]))
PY
  rm -rf $O/r06_s8_prof_$tag
done
grep "^variant\|^bg=" $O/r06_s8_heads_variants.txt | cut -c1-330
