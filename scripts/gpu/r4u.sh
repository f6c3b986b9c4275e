# round 4, call U: e2e after the raw file buffer + pipelined download; whole GPU suite
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4u; mkdir -p $O; cd $R
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $O/tests.log 2>&1; rc=$?; tail -n 3 $O/tests.log
[ $rc -eq 0 ] || exit $rc
for v in 1 2 3; do
  timeout -k 10 400 python3 bench.py --steps 3 --warmup 1 > $O/bench_$v.json 2> $O/bench_$v.err || exit 1
  python3 -c "import json; d=json.loads([l for l in open('$O/bench_$v.json') if l.startswith('{')][-1]); e=d['e2e']; print('$v', d['value'], e['value'], e['equal_to_cpu_baseline'], e['split_ms'], e['sample'][-60:])"
done
