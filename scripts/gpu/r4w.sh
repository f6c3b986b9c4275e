# round 4, call W: resident multi-device slices serving get_penalty; the whole parity + fullsize(config1) suites
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4w; mkdir -p $O; cd $R
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q > $O/tests.log 2>&1; rc=$?; echo "parity rc=$rc"; tail -n 12 $O/tests.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 600 python3 -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "config1" > $O/tests2.log 2>&1; echo "config1 rc=$?"; tail -n 3 $O/tests2.log
