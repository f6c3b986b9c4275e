# round 5, call AD: GPU suite (with the streaming ingest's buffer-route test) and fuzz sets on the library with the page-locked ingest
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5ad; mkdir -p $O; cd $R
timeout -k 10 540 python3 -m pytest tests -q -m gpu -x > $O/tests.log 2>&1; rc=$?; echo "suite rc=$rc"; tail -n 4 $O/tests.log; [ $rc = 0 ] || exit 1
bash scripts/gpu/fuzz.sh r5ad/fuzz 150 && bash scripts/gpu/fuzz.sh r5ad/fuzz_gz 100 gz
