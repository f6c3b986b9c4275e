# round 3, call g: why does the sharded path differ at 15k?  + golden checksums of the configs[4] slice + remaining tests
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3g; mkdir -p $O; cd $R
line() { tag=$1; shift; env "$@" timeout -k 10 300 python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline $EXTRA > $O/$tag.json 2>$O/$tag.err; python3 -c "
import json; d=json.load(open('$O/$tag.json')); print('$tag', d['ms_per_step'], d['counts'], d['checksums'], d.get('parity'))" || tail -n 5 $O/$tag.err; }
EXTRA=""; line direct A=1; line dist1 SEQWIN_BENCH_FORCE_DIST=1; line dist1_legacy SEQWIN_BENCH_FORCE_DIST=1 SEQWIN_AMD_NO_PACKED_EDGES=1; line rccl1 SEQWIN_BENCH_FORCE_DIST=1 SEQWIN_DIST_FORCE_COLLECTIVES=1
EXTRA="--workload salmonella500"; line salm_direct A=1; line salm_dist1 SEQWIN_BENCH_FORCE_DIST=1
cp tests/golden/bench_checksums.json $O/bench_checksums_before.json
for k in 15 19 31; do timeout -k 10 300 python3 bench.py --workload random100k -k $k --steps 2 --warmup 1 --no-cpu-baseline --write-golden > $O/rand_k$k.json 2> $O/rand_k$k.err; python3 -c "
import json; d=json.load(open('$O/rand_k$k.json')); print('random100k k$k', d['value'], d['ms_per_step'], d['stages_ms'], d['counts'])"; done
cp tests/golden/bench_checksums.json $O/bench_checksums.json
timeout -k 10 900 python3 -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py -m gpu -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -n 8 $O/tests.log
