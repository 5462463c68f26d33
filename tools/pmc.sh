# usage: bash tools/pmc.sh <tag> "<counters>" [env...]   -> gpurun_out/pmc_<tag>/ and gpurun_out/pmc_<tag>.json
# One rocprofv3 --pmc pass (counters in their own run, with --kernel-trace only) over a short bench.py run;
# prints / stores per-kernel averages per launch.  BENCH_ARGS adds bench.py arguments.
tag=$1; shift; ctrs=$1; shift
cd /tmp && export TMPDIR=/tmp
env "$@" timeout 400 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 3 --only-timed --repeats 1 $BENCH_ARGS > $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<PY
import csv,glob,collections,json
f=glob.glob('gpurun_out/pmc_$tag/**/*counter_collection.csv', recursive=True)
if not f: print(open('gpurun_out/pmc_$tag.log').read()[-2000:]); raise SystemExit
rows=list(csv.DictReader(open(f[0])))
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for r in rows:
    k=r['Kernel_Name'][:140]; agg[k][r['Counter_Name']]+=float(r['Counter_Value'])
    if r['Counter_Name']==rows[0]['Counter_Name']: cnt[k]+=1
out={}
for k,v in agg.items():
    if 'gs::' in k:
        out[k]={"launches":cnt[k], **{a:round(b/cnt[k]) for a,b in v.items()}}
        if 'composite' in k: print(k[:90], out[k])
json.dump(out, open('gpurun_out/pmc_$tag.json','w'), indent=1)
PY
