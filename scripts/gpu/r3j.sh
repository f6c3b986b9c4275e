export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3j; mkdir -p $O; cd $R
timeout -k 10 200 python3 tests/tools/rccl_self_check.py 149000000 > $O/rccl.log 2>&1; grep "^rep" $O/rccl.log
timeout -k 10 200 python3 tests/tools/rccl_self_check.py 1000000 > $O/rccl_small.log 2>&1; grep "^rep" $O/rccl_small.log
