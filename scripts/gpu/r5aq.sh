# round 5, call AQ: the gzip / ingest tests of the GPU suite on the library with the reworked inflate loop (host code; the CPU suite and
# the differential fuzzer under ASan / UBSan have passed), and the bench's gz leg
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5aq; mkdir -p $O; cd $R
timeout -k 10 300 python3 -m pytest tests -q -m gpu -x -k "gz or ingest or fasta or smoke or dropin or drop_in" > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -n 3 $O/tests.log; [ $rc = 0 ] || exit 1
FUZZ_GZ=1 timeout -k 10 100 python3 tests/tools/fuzz_gpu.py 60 61 > $O/fuzz_gz_host.log 2>&1; echo "fuzz rc=$?"; tail -n 1 $O/fuzz_gz_host.log
