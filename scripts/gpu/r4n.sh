# round 4, call N: the whole GPU suite + the default bench with the own pair sort as the default
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4n; mkdir -p $O; cd $R
timeout -k 10 1100 python3 -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -n 8 $O/tests.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 600 python3 bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -n 3 $O/bench.err
python3 -c "
import json; d=json.load(open('$O/bench.json')); print(d['value'], d['ms_per_step'], d['stages_ms'], d.get('parity'), d['first_build_ms']); print(d['cpu_baseline']['value'], d['e2e']['value'], d['e2e']['split_ms'])"
