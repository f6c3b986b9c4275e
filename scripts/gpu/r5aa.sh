# round 5, call AA: the parsers' word buffers page-locked (DMA from where the packer wrote) and the parsers held within a window of the
# sink: GPU suite, the three forms side by side (tests/tools/e2e_ingest_ab.py, AB_MODES=pinned), the default bench line
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5aa; mkdir -p $O; cd $R
timeout -k 10 480 python3 -m pytest tests -q -m gpu -x > $O/tests.log 2>&1; rc=$?; echo "suite rc=$rc"; tail -n 4 $O/tests.log; [ $rc = 0 ] || exit 1
for nc in 16 32 64; do
  AB_MODES=pinned AB_REPS=3 SEQWIN_AMD_DEBUG_TIMING=1 timeout -k 10 300 python3 tests/tools/e2e_ingest_ab.py 2048 $nc > $O/ab_$nc.txt 2>&1; echo "ab $nc rc=$?"; grep -E "n_cpu=|sink thread" $O/ab_$nc.txt | tail -n 18
done
timeout -k 10 400 python3 bench.py --steps 10 --warmup 2 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; python3 -c "
import json; d=json.load(open('$O/bench_default.json')); print(d['value'], d['ms_per_step'], d['e2e']['value'], d['e2e']['by_n_cpu'], d['e2e']['split_ms'], d['e2e']['gz']['routes'], d['parity']['equal'])"
