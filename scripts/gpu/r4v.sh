# round 4, call V: multi-device edge cases + the multi-device fuzz campaign again
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4v; mkdir -p $O; cd $R
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "multi_device or seqwin_devices" > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -n 8 $O/tests.log
[ $rc -eq 0 ] || exit 1
bash scripts/gpu/fuzz.sh r4v 200 multi
