R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
bash tools/ab.sh "--slots 1" "--batch 4 --slots 2 --join-every 2" "--batch 2 --slots 4 --join-every 4" "--slots 1" "--batch 4 --slots 2 --join-every 2" "--batch 4 --slots 2" \
  "--path heads --slots 1" "--path heads --batch 4 --slots 2 --join-every 2" "--path heads --batch 2 --slots 4 --join-every 4" "--config cfg4 --slots 1" "--config cfg4 --batch 4 --slots 2 --join-every 2" > /dev/null
cp $O/ab.log $O/j_ab_join.txt
