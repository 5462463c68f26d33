# usage: bash tools/pmc.sh <tag> "<counters>" [env...]   -> gpurun_out/pmc_<tag>/
tag=$1; shift; ctrs=$1; shift
cd /tmp && export TMPDIR=/tmp
env "$@" timeout 300 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline $BENCH_ARGS > $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<PY
import csv,glob,collections
f=glob.glob('gpurun_out/pmc_$tag/*counter_collection.csv')
if not f: print(open('gpurun_out/pmc_$tag.log').read()[-2000:]); raise SystemExit
rows=list(csv.DictReader(open(f[0])))
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for r in rows:
    k=r['Kernel_Name'][:60]; agg[k][r['Counter_Name']]+=float(r['Counter_Value'])
    if r['Counter_Name']==rows[0]['Counter_Name']: cnt[k]+=1
for k,v in agg.items():
    if 'gs::' in k: print(k, cnt[k], {a:round(b/cnt[k]) for a,b in v.items()})
PY
