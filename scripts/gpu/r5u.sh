# round 5, call S: the GPU suite and the default bench line on the library whose host ingest maps plain files (mmap + MAP_POPULATE)
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5u; mkdir -p $O; cd $R
timeout -k 10 480 python3 -m pytest tests -q -m gpu -x > $O/tests.log 2>&1; echo "suite rc=$?"; tail -n 4 $O/tests.log
timeout -k 10 300 python3 bench.py --steps 10 --warmup 2 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; python3 -c "
import json; d=json.load(open('$O/bench_default.json')); print(d['value'], d['ms_per_step'], d['cpu_baseline']['value'], d['cpu_baseline'].get('all_cores'), d['e2e']['value'], d['e2e']['by_n_cpu'], d['e2e']['split_ms'], d['e2e']['gz']['routes'], d['parity']['equal'], d['parity']['n1_checksums_equal'])"
