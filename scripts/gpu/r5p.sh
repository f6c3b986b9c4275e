# round 5, call P: the 512-genome set (configs[1] stand-in) pinned to the compiled reference at w = 200 and at w = 10 (4.5e8 occurrences)
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5p; mkdir -p $O; cd $R
for w in 200 10; do
  timeout -k 10 560 python3 scripts/pin_fullsize_ref.py --workload salmonella500 -k 21 -w $w --out $O/pin_salmonella500_w$w.json > $O/pin_salmonella500_w$w.log 2>&1
  echo "pin w=$w rc=$?"; tail -n 4 $O/pin_salmonella500_w$w.log
done
