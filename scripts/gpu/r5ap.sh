# round 5, call AP: the default bench line of the round's last library (after the pool's hand-over change), for profiles/
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5ap; mkdir -p $O; cd $R
timeout -k 10 400 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; python3 -c "
import json; d=json.load(open('$O/bench_default.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['cpu_baseline']['value'], d['e2e']['value'], d['e2e']['first_call'], d['e2e']['by_n_cpu'], d['e2e']['gz']['routes']['host']['Gbp_per_s'], d['parity']['equal'], d['parity']['n1_checksums_equal'])"
