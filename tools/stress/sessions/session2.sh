cd $GRAFT_REPO_ROOT
out=gpurun_out/r2b; mkdir -p $out
export TMPDIR=/tmp
echo "== rocprof kernel stats of the driver's bench command"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-latency > $GRAFT_REPO_ROOT/$out/bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/$out/bench_under_rocprof.err)
find $out/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'head -14 {} | cut -c1-200'
echo "== PMC passes"
bash tools/pmc.sh r2_fetch "FETCH_SIZE"
bash tools/pmc.sh r2_write "WRITE_SIZE"
bash tools/pmc.sh r2_sq "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES"
bash tools/pmc.sh r2_sq2 "SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE"
echo "== stress experiments on the plain build (hipcc's own hazard padding only)"
P="LD_LIBRARY_PATH=gsgen_amd/lib_alt/plain"
env $P python tools/stress/run_matrix.py --out $out/plain_base.jsonl --procs 200 --budget-s 75 --variants mfma2 --batches 0 --Cs 1 --orders test-first,ref-first --launches 3 --extra "--dump-bad 24"
env $P python tools/stress/run_matrix.py --out $out/plain_poison_lds.jsonl --procs 200 --budget-s 45 --variants mfma2 --batches 0 --Cs 1 --orders test-first --launches 3 --extra "--poison --dump-bad 24"
env $P python tools/stress/run_matrix.py --out $out/plain_poison_regs.jsonl --procs 200 --budget-s 45 --variants mfma2 --batches 0 --Cs 1 --orders test-first --launches 3 --extra "--poison-regs --dump-bad 24"
python tools/stress/run_matrix.py --out $out/vec_poison.jsonl --procs 200 --budget-s 30 --variants vec --batches 0,3 --Cs 1,4 --orders test-first --launches 3 --extra "--poison --poison-regs"
python - <<PY
import json,glob
for f in sorted(glob.glob("$out/*.jsonl")):
    for l in open(f):
        d=json.loads(l)
        if d.get("bad_launches") or d.get("rc") not in (0,): print(f.split("/")[-1], json.dumps(d)[:1500])
PY
