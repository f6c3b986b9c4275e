# round 5, call AO: pool blocks that change hands between threads are handed over on the host (api.hip: dev_alloc) -- GPU suite, the
# multi-device fuzz set, and the timed path of the default bench (single thread: nothing should change)
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5ao; mkdir -p $O; cd $R
timeout -k 10 560 python3 -m pytest tests -q -m gpu -x > $O/tests.log 2>&1; rc=$?; echo "suite rc=$rc"; tail -n 4 $O/tests.log; [ $rc = 0 ] || exit 1
timeout -k 10 200 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_15k.json 2> $O/bench_15k.err; echo "bench rc=$?"; python3 -c "
import json; d=json.load(open('$O/bench_15k.json')); print('15k', d['value'], d['ms_per_step'], d['parity'].get('n1_checksums_equal'))"
bash scripts/gpu/fuzz.sh r5ao/fuzz_multi 140 multi
