# round 4, call Y: slice builds whose node sort reads the received rows itself (no k_rows_to_pay): tests, forced-dist bench A/B, fuzz
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4y; mkdir -p $O; cd $R
timeout -k 10 900 python3 -m pytest tests/test_gpu_dist.py tests/test_gpu_fullsize.py tests/test_gpu_parity.py -x -q -m gpu -k "dist or sharded or routed or multi_device or config3 or config4" > $O/tests.log 2>&1; rc=$?; tail -n 3 $O/tests.log
[ $rc -eq 0 ] || exit $rc
for v in copy rows copy rows; do
  if [ $v = copy ]; then export SEQWIN_AMD_ORDER=copy; else unset SEQWIN_AMD_ORDER; fi
  SEQWIN_BENCH_FORCE_DIST=1 timeout -k 10 300 python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline > $O/bench_$v.json 2> $O/bench_$v.err || exit 1
  python3 -c "import json; d=json.loads([l for l in open('$O/bench_$v.json') if l.startswith('{')][-1]); print('$v', d['value'], d['ms_per_step'], d['dist']['phases_ms_max_over_ranks'], d['parity'])"
done
unset SEQWIN_AMD_ORDER
bash scripts/gpu/fuzz.sh r4y ${1:-150}
