# round 5, call M: the pin-script test, then a longer soak of the fuzzer (default set and stage set, 300 s each)
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5m; mkdir -p $O; cd $R
timeout -k 10 300 python3 -m pytest tests/test_gpu_fullsize.py -q -m gpu -x -k "pin_script" > $O/pin_test.log 2>&1; echo "pin test rc=$?"; tail -n 3 $O/pin_test.log
bash scripts/gpu/fuzz.sh r5m/fuzz_default 300 && bash scripts/gpu/fuzz.sh r5m/fuzz_stage 300 stage
