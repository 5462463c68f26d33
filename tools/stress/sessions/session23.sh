cd $GRAFT_REPO_ROOT
out=gpurun_out/r2w; mkdir -p $out
export TMPDIR=/tmp
for cfg in "--res 512 --batch 4" "--res 800 --batch 8"; do
echo "== heads path $cfg"
tag=$(echo $cfg | tr -d ' -')
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/prof_$tag -o bb -- python $GRAFT_REPO_ROOT/tools/bench_batch.py --no-stats --heads --steps 100 $cfg 2>/dev/null | tail -1)
python - <<PY
import csv
for r in list(csv.reader(open("$out/prof_$tag/bb_kernel_stats.csv")))[1:14]:
    print(r[0][:84].ljust(84), r[1], "%.1f us"%(float(r[3])/1e3), r[4])
PY
done
