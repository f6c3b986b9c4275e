# round 4, call S: node sort's first pass reading the sketch stage (SEQWIN_AMD_ORDER=stage, default from 2^26 occurrences on) against k_order's copy
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4s; mkdir -p $O; cd $R
timeout -k 10 900 python3 -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "large_config_branches or config2_full" > $O/tests_knobs.log 2>&1; rc=$?; tail -n 3 $O/tests_knobs.log
[ $rc -eq 0 ] || exit $rc
for i in 1 2 3; do
  for v in copy stage; do
    SEQWIN_AMD_ORDER=$v timeout -k 10 300 python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline > $O/bench_$v.json 2> $O/bench_$v.err || exit 1
    python3 -c "import json,sys; d=json.loads([l for l in open('$O/bench_$v.json') if l.startswith('{')][-1]); print('$v', d['value'], d['ms_per_step'], d['stages_ms'], d.get('parity'))"
  done
done
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stage -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline > $O/prof_stage.log 2>&1
f=$(ls $O/prof_stage/*/*kernel_stats.csv | tail -n 1); python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]: print(r['Name'][:70], r['Calls'], round(float(r['AverageNs'])/1e3,1), r['Percentage'])
PY
