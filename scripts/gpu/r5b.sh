# round 5, call B: full-size pins to the compiled reference (15 000 genomes; one GPU's share of random100k at k = 19), the
# order-guard tests (fault injection) and the whole GPU suite on the new library
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5b; mkdir -p $O; cd $R
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "order_guard" > $O/guard.log 2>&1; echo "guard rc=$?"; tail -n 5 $O/guard.log
timeout -k 10 500 python3 scripts/pin_fullsize_ref.py --workload bacteria15k -k 21 -w 200 --out $O/pin_bacteria15k.json > $O/pin_bacteria15k.log 2>&1
echo "pin15k rc=$?"; tail -n 6 $O/pin_bacteria15k.log
timeout -k 10 300 python3 scripts/pin_fullsize_ref.py --workload random100k -k 19 -w 200 --genomes 2500 --out $O/pin_random100k_k19_probe.json > $O/pin_random100k_k19_probe.log 2>&1
echo "probe rc=$?"; tail -n 4 $O/pin_random100k_k19_probe.log
timeout -k 10 420 python3 -m pytest tests -q -m gpu -x > $O/tests.log 2>&1; echo "suite rc=$?"; tail -n 8 $O/tests.log
