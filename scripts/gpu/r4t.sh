# round 4, call T: final state -- the whole GPU suite, then the profile set
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4t; mkdir -p $O; cd $R
timeout -k 10 1100 python3 -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -n 6 $O/tests.log
[ $rc -eq 0 ] || exit 1
python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -n 1 $O/smoke.log
