# round 5, call J: one GPU's share of random100k at k = 15 and k = 31 pinned to the compiled reference (k = 19: call E)
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5j; mkdir -p $O; cd $R
for k in 15 31; do
  timeout -k 10 560 python3 scripts/pin_fullsize_ref.py --workload random100k -k $k -w 200 --size-from "tests/golden/bench_checksums_ref.json#random100k/k19/w200" --out $O/pin_random100k_k$k.json > $O/pin_random100k_k$k.log 2>&1
  echo "pin k=$k rc=$?"; tail -n 4 $O/pin_random100k_k$k.log
done
