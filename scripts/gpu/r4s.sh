# round 4, call S: first build with / without the pool reservation; parity subset; default bench
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4s; mkdir -p $O; cd $R
for v in 1 0 1 0; do
  if [ $v = 0 ]; then export SEQWIN_AMD_NO_POOL_RESERVE=1; else unset SEQWIN_AMD_NO_POOL_RESERVE; fi
  timeout -k 10 300 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/b_$v.json 2> $O/b_$v.err; python3 -c "
import json; d=json.load(open('$O/b_$v.json')); print('reserve=$v', d['value'], d['ms_per_step'], 'first', d['first_build_ms'], 'plan', d['plan_ms'], d.get('parity'))"; done
unset SEQWIN_AMD_NO_POOL_RESERVE
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "two_host_threads or resident or low_memory or fuzz or full_size" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -n 3 $O/tests.log
