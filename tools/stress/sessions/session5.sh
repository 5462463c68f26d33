cd $GRAFT_REPO_ROOT
out=gpurun_out/r2e; mkdir -p $out
export TMPDIR=/tmp
echo "== pytest -m gpu"
timeout 2400 python -m pytest tests -m gpu -q > $out/pytest.log 2>&1; echo "pytest rc=$? $(tail -1 $out/pytest.log)"
grep -n "^FAILED\|^ERROR" $out/pytest.log | head -20
echo "== A/B bench"
for v in "X=1" "GSGEN_HIP_LIB=$GRAFT_REPO_ROOT/gsgen_amd/lib_alt/nopairskip/libgsgen_hip.so" "X=2"; do
  env $v timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > "$out/bench_${v%%=*}${v##*/}.json" 2> $out/bench.err
  python -c "
import json,sys
d=json.load(open(sys.argv[1])); r=d['roofline']; o=d.get('one_render_in_flight',{})
print(sys.argv[2][:40], round(d['value'],1), r['kernel'], 'bwd', round(r['avg_launch_ms'],3), 'alone', round(r['alone_launch_ms'],3), 'fwd', round(r['fwd_launch_ms'],3), 'alone fwd', round(r['alone_fwd_launch_ms'],3), 'one', round(o.get('value',0),1))" "$out/bench_${v%%=*}${v##*/}.json" "$v" || tail -5 $out/bench.err
done
echo "== PMC"
bash tools/pmc.sh r2e_sq "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES"
