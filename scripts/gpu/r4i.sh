# round 4, call I: 4-byte aggregates + one inclusive probe per step
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4i; mkdir -p $O; cd $R
T="timeout -k 10 200 python3 tests/tools/sort_time.py 745"
timeout -k 10 300 python3 tests/tools/sort_check.py 80 > $O/sort_check.log 2>&1; rc=$?; echo "sort_check rc=$rc"; grep -c OK $O/sort_check.log; grep -v "OK\|amdgpu" $O/sort_check.log | tail -n 5
[ $rc -eq 0 ] || exit 1
$T 45 > $O/st_45.log 2>&1; grep bits= $O/st_45.log
$T 54 > $O/st_54.log 2>&1; grep bits= $O/st_54.log
SEQWIN_AMD_STAMPS=1 SEQWIN_AMD_LIB=$R/ab/libseqwin_hip_rsst.so $T 45 > $O/stamps.log 2>&1; grep "rs stamps" $O/stamps.log | tail -n 2
timeout -k 10 300 python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -n 3 $O/bench.err
python3 -c "
import json; d=json.load(open('$O/bench.json')); print(d['value'], d['ms_per_step'], d['stages_ms'], d.get('parity'))"
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "sort or fuzz or synthetic" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -n 3 $O/tests.log
