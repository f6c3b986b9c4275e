# round 4, call F: pair-table warm-up of the sketch kernel + reordered radix pass: parity, sort checks, timing, stamps, bench
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4f; mkdir -p $O; cd $R
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q > $O/tests_parity.log 2>&1; rc=$?; echo "parity rc=$rc"; tail -n 6 $O/tests_parity.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python3 tests/tools/sort_check.py 80 > $O/sort_check.log 2>&1; rc=$?; echo "sort_check rc=$rc"; grep -c OK $O/sort_check.log; grep -v "OK\|amdgpu" $O/sort_check.log | tail -n 5
[ $rc -eq 0 ] || exit 1
T="timeout -k 10 200 python3 tests/tools/sort_time.py 745"
$T 45 > $O/st_45.log 2>&1; grep bits= $O/st_45.log
$T 54 > $O/st_54.log 2>&1; grep bits= $O/st_54.log
SEQWIN_AMD_RADIX_SHAPE=1024x8 $T 48 > $O/st_1024x8.log 2>&1; grep bits= $O/st_1024x8.log
SEQWIN_AMD_STAMPS=1 SEQWIN_AMD_LIB=$R/ab/libseqwin_hip_rsst.so $T 45 > $O/stamps.log 2>&1; grep "rs stamps" $O/stamps.log | tail -n 2
timeout -k 10 300 python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -n 3 $O/bench.err
python3 -c "
import json; d=json.load(open('$O/bench.json')); print(d['value'], d['ms_per_step'], d['stages_ms'], d.get('parity'))"
SEQWIN_AMD_STAMPS=1 SEQWIN_AMD_LIB=$R/ab/libseqwin_hip_skst.so timeout -k 10 300 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_skst.json 2> $O/bench_skst.err; grep -i stamps $O/bench_skst.err | tail -n 2
for wl in "salmonella500" "random100k -k 19" "random100k -k 31" "salmonella500 -w 10"; do timeout -k 10 300 python3 bench.py --workload $wl --steps 4 --warmup 1 --no-cpu-baseline > $O/b.json 2>> $O/b.err; python3 -c "
import json; d=json.load(open('$O/b.json')); print('$wl', d['value'], d['ms_per_step'], d['stages_ms'], d.get('parity'))"; done
