# round 5, call AI: the final library pinned again, element for element, to the compiled reference at full size -- the 15 000 genomes
# and one GPU's share of random100k at k = 15 (the configuration whose adjacency kernel changed at the end of the round)
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5ai; mkdir -p $O; cd $R
timeout -k 10 520 python3 scripts/pin_fullsize_ref.py --workload bacteria15k -k 21 -w 200 --out $O/pin_bacteria15k.json > $O/pin_bacteria15k.log 2>&1
echo "pin15k rc=$?"; tail -n 6 $O/pin_bacteria15k.log
timeout -k 10 520 python3 scripts/pin_fullsize_ref.py --workload random100k -k 15 -w 200 --size-from "tests/golden/bench_checksums_ref.json#random100k/k19/w200" --out $O/pin_random100k_k15.json > $O/pin_random100k_k15.log 2>&1
echo "pin r100k k15 rc=$?"; tail -n 6 $O/pin_random100k_k15.log
