cd $GRAFT_REPO_ROOT
for a in "--slots 3" "--slots 2"; do
  timeout 20 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-latency --repeats 5 $a 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(sys.argv[1], '|', round(d['value'],1), 'ms/step', round(d['ms_per_step'],3), 'bwd', round(r['avg_launch_ms'],3), 'fwd', round(r['fwd_launch_ms'],3), 'exact', round(d['exact_basis']['value'],1))" "$a"
done
