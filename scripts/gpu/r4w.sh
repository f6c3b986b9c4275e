# round 4, call W: kernel averages of a 4-step bench under rocprofv3 (after the knob tests of the full-size file)
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4w; mkdir -p $O; cd $R
timeout -k 10 900 python3 -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "large_config_branches or config2_full" > $O/tests_knobs.log 2>&1; rc=$?; tail -n 3 $O/tests_knobs.log
[ $rc -eq 0 ] || exit $rc
for v in a b; do
  cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$v -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline > $O/prof_$v.log 2>&1
  f=$(ls -t $O/prof_$v/*/*kernel_stats.csv | head -n 1); python3 - "$f" $v <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r['TotalDurationNs']) for r in rows if 'k_synth' not in r['Name'] and 'k_checksum' not in r['Name'])/5e6
sel=[(r['Name'].split('(anonymous namespace)::')[-1][:24], round(float(r['AverageNs'])/1e3,1)) for r in rows if any(k in r['Name'] for k in ('k_unsort','k_nodes','pair_pass','k_pen_bits'))]
print(sys.argv[2], round(tot,2), sel)
PY
  grep "^{" $O/prof_$v.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['stages_ms'], d['parity'])"
done
