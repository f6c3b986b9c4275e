# round 4, call T: whole GPU suite, smoke(), then a fuzz campaign (five or four processes): bash scripts/gpu/r4t.sh <seconds> [gz|multi|stage]
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4t; mkdir -p $O; cd $R
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $O/tests.log 2>&1; rc=$?; tail -n 3 $O/tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" > $O/smoke.log 2>&1; rc=$?; tail -n 2 $O/smoke.log
[ $rc -eq 0 ] || exit $rc
bash scripts/gpu/fuzz.sh r4t ${1:-200} $2
