# round 4, call U: the count-only path on the GPU (single process, gloo x2/x3, RCCL world 1), then the final profile set
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4u; mkdir -p $O; cd $R
timeout -k 10 600 python3 -m pytest tests/test_gpu_dist.py -m gpu -x -q -k "count_only or rccl" > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -n 8 $O/tests.log
[ $rc -eq 0 ] || exit 1
bash scripts/gpu/prof.sh r4u
