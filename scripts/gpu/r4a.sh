# round 4, call A: atomic-rank radix passes (check + timing), the new configs[4] tests, the dist suite, a sharded bench line
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4a; mkdir -p $O; cd $R
timeout -k 10 300 python3 tests/tools/sort_check.py 50 > $O/sort_check.log 2>&1; echo "sort_check rc=$?"; tail -n 12 $O/sort_check.log
for m in ballot atomic; do SEQWIN_AMD_RADIX_RANK=$m timeout -k 10 200 python3 tests/tools/sort_time.py 745 54 >> $O/sort_time.log 2>&1; SEQWIN_AMD_RADIX_RANK=$m timeout -k 10 200 python3 tests/tools/sort_time.py 745 45 >> $O/sort_time.log 2>&1; done
cat $O/sort_time.log
timeout -k 10 900 python3 -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "config4_share or config4_routed" > $O/tests_c4.log 2>&1; rc1=$?; echo "c4 rc=$rc1"; tail -n 15 $O/tests_c4.log
[ $rc1 -eq 0 ] || exit 1
timeout -k 10 600 python3 -m pytest tests/test_gpu_dist.py -m gpu -x -q > $O/tests_dist.log 2>&1; rc2=$?; echo "dist rc=$rc2"; tail -n 5 $O/tests_dist.log
[ $rc2 -eq 0 ] || exit 1
SEQWIN_BENCH_FORCE_DIST=1 SEQWIN_DIST_FORCE_COLLECTIVES=1 timeout -k 10 300 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_dist1.json 2> $O/bench_dist1.err; echo "bench dist rc=$?"
python3 -c "
import json; d=json.load(open('$O/bench_dist1.json')); print(d['value'], d['ms_per_step'], d['dist'], d.get('parity'))"
timeout -k 10 300 python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python3 -c "
import json; d=json.load(open('$O/bench.json')); print(d['value'], d['ms_per_step'], d['stages_ms'], d.get('parity'))"
