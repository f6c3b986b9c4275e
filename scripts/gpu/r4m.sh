# round 4, call M: pair passes (7168-element tiles, one staging round) checked and timed for real (SEQWIN_AMD_PAIR_SORT)
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4m; mkdir -p $O; cd $R
timeout -k 10 400 python3 tests/tools/pair_sort_check.py 30 745 > $O/pair_sort_check.log 2>&1; rc=$?; echo "pair_sort_check rc=$rc"; grep -v "amdgpu.ids\|^  \.\." $O/pair_sort_check.log | tail -n 16
