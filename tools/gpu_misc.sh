R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
for g in "" "--graph"; do timeout 300 python tools/bench_step.py $g 2>/dev/null | tail -1; done > $O/m_bench_step.txt; cat $O/m_bench_step.txt
timeout 300 python tools/host_profile.py 2>/dev/null | tail -3 > $O/m_host_profile.txt; cat $O/m_host_profile.txt
PULL=GSGEN_BIN_PUSH_MIN_WORKGROUPS=100000; PUSH=GSGEN_BIN_PUSH_MIN_WORKGROUPS=1
bash tools/ab.sh "$PULL --batch 1" "$PUSH --batch 1" "$PULL --batch 2" "$PUSH --batch 2" "$PULL --batch 4" "$PUSH --batch 4" "$PULL --batch 1 --slots 1" "$PUSH --batch 1 --slots 1" "$PULL --batch 2 --slots 1" "$PUSH --batch 2 --slots 1" > /dev/null
cp $O/ab.log $O/m_ab_push_threshold.txt
