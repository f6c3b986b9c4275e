# round 5, call AR: the GPU suite on the round's last library
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5ar; mkdir -p $O; cd $R
timeout -k 10 360 python3 -m pytest tests -q -m gpu -x > $O/tests.log 2>&1; echo "suite rc=$?"; tail -n 4 $O/tests.log
