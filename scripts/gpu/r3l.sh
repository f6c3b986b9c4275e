# round 3, call l: whole GPU suite (new full-size tests included) + the default bench line
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3l; mkdir -p $O; cd $R
timeout -k 10 1000 python3 -m pytest tests -m gpu -q --durations=12 > $O/tests.log 2>&1; echo "tests rc=$?"; tail -n 22 $O/tests.log
timeout -k 10 600 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; python3 -c "
import json; d=json.load(open('$O/bench_default.json')); print(d['value'], d['ms_per_step'], d['stages_ms'], 'plan', d['plan_ms'], 'first', d['first_build_ms']); print(d['roofline']['kernel'], d['roofline']['frac'], d['parity']); print(d['cpu_baseline']); print(d['e2e'])"
python3 __graft_entry__.py smoke 2>&1 | tail -n 2
