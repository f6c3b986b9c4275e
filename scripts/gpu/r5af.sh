# round 5, call AF: sw_get_penalty returns 16 of a node's 40 bytes (packed on the device, put into place by the download's copiers):
# its tests, the download / multi / drop-in tests, then the default bench line (e2e split: get_penalty_wall_ms)
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5af; mkdir -p $O; cd $R
timeout -k 10 540 python3 -m pytest tests -q -m gpu -x > $O/tests.log 2>&1; rc=$?; echo "suite rc=$rc"; tail -n 4 $O/tests.log; [ $rc = 0 ] || exit 1
timeout -k 10 400 python3 bench.py --steps 10 --warmup 2 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; python3 -c "
import json; d=json.load(open('$O/bench_default.json')); print(d['value'], d['ms_per_step'], d['e2e']['value'], d['e2e']['first_call'], d['e2e']['by_n_cpu'], d['e2e']['split_ms'], d['e2e']['equal_to_cpu_baseline'], d['parity']['equal'])"
SEQWIN_AMD_PENALTY_WHOLE_NODES=1 SEQWIN_BENCH_E2E_NCPU=32,128 timeout -k 10 400 python3 bench.py --steps 3 --warmup 1 > $O/bench_whole.json 2> $O/bench_whole.err; echo "bench whole rc=$?"; python3 -c "
import json; d=json.load(open('$O/bench_whole.json')); print('whole nodes:', d['e2e']['value'], d['e2e']['by_n_cpu'], d['e2e']['split_ms'])"
bash scripts/gpu/fuzz.sh r5af/fuzz 120
