# round 4, call R: the new full-size tests (multi-device at configs[1] size, knob rows of the new sort paths)
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4r; mkdir -p $O; cd $R
timeout -k 10 1100 python3 -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "config1 or large_config" > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -n 8 $O/tests.log
