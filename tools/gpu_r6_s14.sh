# round 6, session 14: the sort launch -- short lists four to a workgroup (a wavefront each); threshold 256 (in-tree) / 512, against the previous build
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_overflow.py -m gpu -x -q -p no:cacheprovider > $O/r06_s14_gpu_tests.log 2>&1; echo "pytest rc=$?"; tail -3 $O/r06_s14_gpu_tests.log
cd /tmp; export TMPDIR=/tmp
: > $O/r06_s14_sort_alone.txt
for v in sort_prev - sort_512; do
  if [ "$v" = "-" ]; then envs="X=1"; else envs="GSGEN_HIP_LIB=$R/gsgen_amd/lib_alt/$v.so"; fi
  rm -rf /tmp/prof_$v; env $envs timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$v -o t -- python $R/tools/prof_heads_variants.py 0 0 0 2>&1 | grep "ms per step" | sed "s/^/$v: /" >> $O/r06_s14_sort_alone.txt
  f=$(find /tmp/prof_$v -name "*kernel_stats.csv" | head -1)
  python - "$f" "$v" >> $O/r06_s14_sort_alone.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n = r["Name"]
    if any(k in n for k in ("sort", "bin_push", "scan", "project", "composite")):
        print(f"  {sys.argv[2]:10s} {float(r['AverageNs'])/1e3:9.1f} us x {r['Calls']:>4s}  {n[:90]}")
PY
done
cat $O/r06_s14_sort_alone.txt
cd $R
AB_ARGS="--no-other-configs" bash tools/ab_heads.sh r06_s14 2 $R/gsgen_amd/lib_alt/sort_prev.so - $R/gsgen_amd/lib_alt/sort_512.so
