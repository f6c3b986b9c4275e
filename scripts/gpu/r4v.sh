# round 4, call V: A/B of one knob (separate = the knob set), three alternating runs
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4v; mkdir -p $O; cd $R
timeout -k 10 900 python3 -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "large_config_branches or config2_full" > $O/tests_knobs.log 2>&1; rc=$?; tail -n 3 $O/tests_knobs.log
[ $rc -eq 0 ] || exit $rc
for i in 1 2 3; do
  for v in separate fused; do
    if [ $v = separate ]; then export SEQWIN_AMD_DESC_LOW=0; else unset SEQWIN_AMD_DESC_LOW; fi
    timeout -k 10 300 python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline > $O/bench_$v.json 2> $O/bench_$v.err || exit 1
    python3 -c "import json,sys; d=json.loads([l for l in open('$O/bench_$v.json') if l.startswith('{')][-1]); print('$v', d['value'], d['ms_per_step'], d['stages_ms'], d.get('parity'))"
  done
done
