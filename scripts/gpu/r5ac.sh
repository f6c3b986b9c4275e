# round 5, call AC: the pool of page-locked blocks grows without anybody waiting for it (first call of a process), smaller window;
# bench's e2e leg reports the first call on its own
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5ac; mkdir -p $O; cd $R
for nc in 32 128; do
  AB_MODES=pinned AB_REPS=2 SEQWIN_AMD_DEBUG_TIMING=1 timeout -k 10 300 python3 tests/tools/e2e_ingest_ab.py 2048 $nc > $O/ab_$nc.txt 2>&1; echo "ab $nc rc=$?"; grep -E "n_cpu=|sink thread" $O/ab_$nc.txt | tail -n 12
done
timeout -k 10 400 python3 bench.py --steps 10 --warmup 2 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; python3 -c "
import json; d=json.load(open('$O/bench_default.json')); print(d['value'], d['ms_per_step'], d['e2e']['value'], d['e2e']['first_call'], d['e2e']['by_n_cpu'], d['e2e']['split_ms'], d['e2e']['gz']['routes'], d['parity']['equal'])"
timeout -k 10 480 python3 -m pytest tests -q -m gpu -x > $O/tests.log 2>&1; echo "tests rc=$?"; tail -n 3 $O/tests.log
