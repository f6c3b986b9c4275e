# round 4, call B: sort checks (keys + pairs, both ranking modes), parity suite, the configs[4] tests, bench A/B of the pair sort
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4c; mkdir -p $O; cd $R
timeout -k 10 300 python3 tests/tools/sort_check.py 80 > $O/sort_check.log 2>&1; echo "sort_check rc=$?"; grep -v amdgpu.ids $O/sort_check.log | tail -n 20
timeout -k 10 400 python3 tests/tools/pair_sort_check.py 30 745 > $O/pair_sort_check.log 2>&1; rc=$?; echo "pair_sort_check rc=$rc"; grep -v amdgpu.ids $O/pair_sort_check.log | grep -v "^  \.\." | tail -n 30
[ $rc -eq 0 ] || exit 1
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q > $O/tests_parity.log 2>&1; rc=$?; echo "parity rc=$rc"; tail -n 5 $O/tests_parity.log
[ $rc -eq 0 ] || exit 1
for v in own rocprim; do SEQWIN_AMD_PAIR_SORT=$v timeout -k 10 300 python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_$v.json 2> $O/bench_$v.err; echo "bench $v rc=$?"
python3 -c "
import json; d=json.load(open('$O/bench_$v.json')); print('$v', d['value'], d['ms_per_step'], d['stages_ms'], d.get('parity'))"; done
timeout -k 10 900 python3 -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "config4_share or config4_routed" > $O/tests_c4.log 2>&1; rc1=$?; echo "c4 rc=$rc1"; tail -n 15 $O/tests_c4.log
[ $rc1 -eq 0 ] || exit 1
timeout -k 10 600 python3 -m pytest tests/test_gpu_dist.py -m gpu -x -q > $O/tests_dist.log 2>&1; rc2=$?; echo "dist rc=$rc2"; tail -n 5 $O/tests_dist.log
