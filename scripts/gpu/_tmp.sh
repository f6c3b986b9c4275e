export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3q; mkdir -p $O; cd $R
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -n 3 $O/tests.log
timeout -k 10 100 python3 tests/tools/fuzz_gpu.py 50 71 > $O/fuzz_default.log 2>&1 &
SEQWIN_AMD_RC=3 SEQWIN_AMD_SLOT_CAP=3 timeout -k 10 100 python3 tests/tools/fuzz_gpu.py 50 72 > $O/fuzz_rc3.log 2>&1 &
wait; tail -n 1 $O/fuzz_default.log $O/fuzz_rc3.log
bash scripts/gpu/ab.sh r3q noemitasm stamps
