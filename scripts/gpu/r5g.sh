# round 5, call G: the host inflate (fast_inflate.hpp) on the GPU box -- ingest tests, default bench with its .gz leg --, and the
# edge sort leaving out 1 / 2 / 3 digits for the in-place repair (thread form up to 2^26 descents) on random100k and on the default set
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5g; mkdir -p $O; cd $R
timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "gz or ingest or upload or fasta" > $O/gz_tests.log 2>&1; echo "gz tests rc=$?"; tail -n 4 $O/gz_tests.log
run() { tag=$1; shift; timeout -k 10 300 python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline "$@" > $O/$tag.json 2>$O/$tag.err; python3 -c "
import json; d=json.load(open('$O/$tag.json')); print('$tag', d['value'], d['ms_per_step'], d['stages_ms'], (d.get('parity') or {}).get('n1_checksums_equal'))"; }
for s in 1 2 3; do SEQWIN_AMD_EDGE_SKIP_PASSES=$s SEQWIN_AMD_DEBUG_EDGE_REPAIR=1 run r100k_k19_skip$s --workload random100k -k 19 --steps 4; grep "edge repair" $O/r100k_k19_skip$s.err | tail -n 1; done
for s in 1 2; do SEQWIN_AMD_EDGE_SKIP_PASSES=$s SEQWIN_AMD_DEBUG_EDGE_REPAIR=1 run b15k_skip$s; grep "edge repair" $O/b15k_skip$s.err | tail -n 1; done
for s in 1 2; do SEQWIN_AMD_EDGE_SKIP_PASSES=$s SEQWIN_AMD_DEBUG_EDGE_REPAIR=1 run w10_skip$s --workload salmonella500 -w 10; grep "edge repair" $O/w10_skip$s.err | tail -n 1; done
timeout -k 10 300 python3 bench.py --steps 10 --warmup 2 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; python3 -c "
import json; d=json.load(open('$O/bench_default.json')); print(d['value'], d['ms_per_step'], d['cpu_baseline'].get('all_cores'), d['e2e']['by_n_cpu'], d['e2e']['gz'], d['parity'])"
