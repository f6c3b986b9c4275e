# round 5, call AJ: the round's last library -- smoke(), the GPU suite, the default bench line
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5aj; mkdir -p $O; cd $R
timeout -k 10 200 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -n 2 $O/smoke.log
timeout -k 10 560 python3 -m pytest tests -q -m gpu -x > $O/tests.log 2>&1; rc=$?; echo "suite rc=$rc"; tail -n 4 $O/tests.log; [ $rc = 0 ] || exit 1
timeout -k 10 400 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; python3 -c "
import json; d=json.load(open('$O/bench_default.json')); print(d['value'], d['ms_per_step'], d['steps'], d['warmup'], d['roofline']['frac'], d['cpu_baseline']['value'], d['e2e']['value'], d['e2e']['first_call'], d['e2e']['by_n_cpu'], d['e2e']['split_ms']['export_ms'], d['e2e']['gz']['routes']['host']['Gbp_per_s'], d['parity']['equal'])"
