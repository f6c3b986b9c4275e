# round 4, call Z: the whole GPU suite with radix.hip's sorts, the stage order and the bucketed unsort forced on every input, however small
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4z; mkdir -p $O; cd $R
SEQWIN_AMD_SORT=own SEQWIN_AMD_UNSORT_DIRECT=4 timeout -k 10 1100 python3 -m pytest tests -q -m gpu > $O/tests_own.log 2>&1; tail -n 15 $O/tests_own.log
