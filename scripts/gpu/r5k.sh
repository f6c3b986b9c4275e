# round 5, call K: the GPU suite on the final library, then fuzz campaigns: every file gzipped (levels 0-9) through the host route
# (fast_inflate.hpp) and through the device route; the default set once more
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5k; mkdir -p $O; cd $R
timeout -k 10 480 python3 -m pytest tests -q -m gpu -x > $O/tests.log 2>&1; echo "suite rc=$?"; tail -n 4 $O/tests.log
T=240
FUZZ_GZ=1 SEQWIN_AMD_DEVICE_INFLATE=0 python3 tests/tools/fuzz_gpu.py $T 51 > $O/fuzz_host_gz.log 2>&1 &
FUZZ_GZ=1 SEQWIN_AMD_DEVICE_INFLATE=0 SEQWIN_AMD_RC=3 SEQWIN_AMD_SLOT_CAP=3 python3 tests/tools/fuzz_gpu.py $T 52 > $O/fuzz_host_gz_rc3.log 2>&1 &
FUZZ_GZ=1 SEQWIN_AMD_DEVICE_INFLATE=1 python3 tests/tools/fuzz_gpu.py $T 53 > $O/fuzz_device_gz.log 2>&1 &
python3 tests/tools/fuzz_gpu.py $T 54 > $O/fuzz_default.log 2>&1 &
SEQWIN_DEVICES=0,0,0 SEQWIN_MULTI_NO_P2P=1 python3 tests/tools/fuzz_gpu.py $T 55 > $O/fuzz_devices3_staged.log 2>&1 &
for i in $(seq 1 40); do sleep 30; echo "t=$((i*30))s"; kill -0 $! 2>/dev/null || break; done
wait
tail -n 1 $O/fuzz_*.log
