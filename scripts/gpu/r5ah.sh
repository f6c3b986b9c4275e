# round 5, call AH: the default bench line of the final library (profiles/r05_bench_default.json) and a fuzz set
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5ah; mkdir -p $O; cd $R
timeout -k 10 400 python3 bench.py --steps 10 --warmup 2 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; python3 -c "
import json; d=json.load(open('$O/bench_default.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['cpu_baseline']['value'], d['e2e']['value'], d['e2e']['first_call'], d['e2e']['by_n_cpu'], d['e2e']['split_ms'], d['e2e']['gz']['routes']['host'], d['e2e']['equal_to_cpu_baseline'], d['parity']['equal'])"
bash scripts/gpu/fuzz.sh r5ah/fuzz 120
