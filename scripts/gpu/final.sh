# final.sh <tag>: the end-of-round set on the final library -- ragged / salmonella lines, the profile set of the default bench, the GPU suite
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O; cd $R
for wl in ragged500 salmonella500; do timeout -k 10 200 python3 bench.py --workload $wl --no-cpu-baseline > $O/bench_$wl.json 2> $O/bench_$wl.err || exit 1; done
python3 -c "
import json
for n in ('ragged500','salmonella500'):
    d=json.load(open('$O/bench_%s.json'%n)); print(n, d['value'], d['ms_per_step'], d['stages_ms'], d.get('tiles'), d.get('parity'))"
bash scripts/gpu/prof.sh $1 > $O/prof.log 2>&1; echo "prof rc=$?"; tail -n 3 $O/prof.log
timeout -k 10 700 python -m pytest tests -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; tail -n 4 $O/tests.log
