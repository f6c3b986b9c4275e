# round 4, call D: multi-device build tests; stamps of the atomic-rank keys pass
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4d; mkdir -p $O; cd $R
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "multi_device or seqwin_devices" > $O/tests_multi.log 2>&1; rc=$?; echo "multi rc=$rc"; tail -n 25 $O/tests_multi.log
for m in atomic ballot; do SEQWIN_AMD_RADIX_RANK=$m SEQWIN_AMD_STAMPS=1 SEQWIN_AMD_LIB=$R/ab/libseqwin_hip_rsst.so timeout -k 10 200 python3 tests/tools/sort_time.py 745 45 > $O/stamps_$m.log 2>&1; grep "rs stamps" $O/stamps_$m.log | tail -n 5; done
