cd $GRAFT_REPO_ROOT
bash tools/stress/gpu_round.sh r2a
out=gpurun_out/r2a
echo "== A/B bench"
for v in "GSGEN_PPL_BWD_SH_BATCH=2" "GSGEN_PPL_BWD_SH_BATCH=1" "GSGEN_BWD_MFMA_BATCH=2"; do
  env $v timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-latency > $out/bench_$v.json 2> $out/bench_$v.err
  python -c "
import json
d=json.load(open('$out/bench_$v.json')); print('$v', round(d['value'],1), d['roofline']['kernel'], round(d['roofline']['avg_launch_ms'],3), 'alone', round(d['roofline']['alone_launch_ms'],3))" || tail -5 $out/bench_$v.err
done
echo "== mfma_first_launch"
for cfg in "--waves 2" "--waves 4" "--waves 2 --vgpr-pad" "--waves 1 --vgpr-pad"; do
  for i in $(seq 1 25); do timeout 60 tools/stress/mfma_first_launch $cfg >> "$out/mfl.jsonl"; done
done
python - <<PY
import json,collections
rows=collections.defaultdict(lambda:[0,0,set()])
for l in open("$out/mfl.jsonl"):
    d=json.loads(l); k=(d["waves"],d["pad"]); rows[k][0]+=1; rows[k][1]+=d["bad_launches"]>0; rows[k][2].add(d["checksum"])
for k,v in rows.items(): print("mfl", k, "procs", v[0], "bad", v[1], "distinct checksums", len(v[2]))
PY
echo "== stress default lib"
python tools/stress/run_matrix.py --out $out/stress_default.jsonl --procs 10 --budget-s 150
echo "== stress short lib"
LD_LIBRARY_PATH=gsgen_amd/lib_alt/short python tools/stress/run_matrix.py --out $out/stress_short.jsonl --procs 40 --budget-s 90 --variants mfma2 --batches 0 --Cs 1,4 --orders test-first
echo "== stress plain lib"
LD_LIBRARY_PATH=gsgen_amd/lib_alt/plain python tools/stress/run_matrix.py --out $out/stress_plain.jsonl --procs 40 --budget-s 90 --variants mfma2,mfma4 --batches 0 --Cs 1,4 --orders test-first
