# round 4, call K: pair passes with 8192-element tiles: check + timing vs rocPRIM; bench with both
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4l; mkdir -p $O; cd $R
timeout -k 10 300 python3 tests/tools/sort_check.py 80 > $O/sort_check.log 2>&1; rc=$?; echo "sort_check rc=$rc"; grep -c OK $O/sort_check.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 400 python3 tests/tools/pair_sort_check.py 30 745 > $O/pair_sort_check.log 2>&1; rc=$?; echo "pair_sort_check rc=$rc"; grep -v "amdgpu.ids\|^  \.\." $O/pair_sort_check.log | tail -n 16
[ $rc -eq 0 ] || exit 1
for v in own rocprim own rocprim; do SEQWIN_AMD_PAIR_SORT=$v timeout -k 10 300 python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_$v.json 2> $O/bench_$v.err; python3 -c "
import json; d=json.load(open('$O/bench_$v.json')); print('$v', d['value'], d['ms_per_step'], d['stages_ms'], d.get('parity'))"; done
