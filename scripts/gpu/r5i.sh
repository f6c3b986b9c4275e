# round 5, call I: (1) the sharded path's cost on one card, round-5 numbers (whole 15k set and one GPU's share at P = 8; without and
# with every collective issued over RCCL at world size 1); (2) the ingest's scaling table with the worker cap lifted; (3) the
# default bench line
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5i; mkdir -p $O; cd $R
run() { tag=$1; shift; timeout -k 10 300 python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline "$@" > $O/$tag.json 2>$O/$tag.err; python3 -c "
import json; d=json.load(open('$O/$tag.json')); print('$tag', d['value'], d['ms_per_step'], d['stages_ms'], (d.get('dist') or {}).get('phases_ms_max_over_ranks'), (d.get('parity') or {}).get('n1_checksums_equal'))"; }
run direct15k
SEQWIN_BENCH_FORCE_DIST=1 run dist15k
SEQWIN_BENCH_FORCE_DIST=1 SEQWIN_DIST_FORCE_COLLECTIVES=1 run dist15k_coll
run direct1875 --genomes 1875
SEQWIN_BENCH_FORCE_DIST=1 run dist1875 --genomes 1875
SEQWIN_BENCH_FORCE_DIST=1 SEQWIN_DIST_FORCE_COLLECTIVES=1 run dist1875_coll --genomes 1875
SEQWIN_AMD_INGEST_WORKERS_MAX=1024 SEQWIN_BENCH_E2E_NCPU=8,16,32,64,128,256 timeout -k 10 400 python3 bench.py --steps 3 --warmup 1 > $O/bench_nocap.json 2> $O/bench_nocap.err; python3 -c "
import json; d=json.load(open('$O/bench_nocap.json')); print('cap lifted', d['e2e']['by_n_cpu'], d['e2e']['gz']['routes']['host'])"
timeout -k 10 300 python3 bench.py --steps 10 --warmup 2 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; python3 -c "
import json; d=json.load(open('$O/bench_default.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['cpu_baseline']['value'], d['cpu_baseline'].get('all_cores'), d['e2e']['by_n_cpu'], d['e2e']['gz']['value'], d['parity'], d['dist']['distinct_gpus'], d['dist']['ranks'])"
