export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3k; mkdir -p $O; cd $R
timeout -k 10 200 python3 tests/tools/rccl_self_check.py 149000000 > $O/rccl.log 2>&1; grep "rep" $O/rccl.log
timeout -k 10 200 python3 tests/tools/rccl_self_check.py 600000000 > $O/rccl_big.log 2>&1; grep "rep" $O/rccl_big.log
line() { tag=$1; shift; env "$@" timeout -k 10 300 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline $EXTRA 2>$O/$tag.err | grep "^{" > $O/$tag.json; python3 -c "
import json; d=json.load(open('$O/$tag.json')); print('$tag', d['ms_per_step'], d['counts'], d['checksums'], d.get('parity'), {k: round(v,1) for k,v in d['stages_ms'].items() if 'wall' in k})" || tail -n 5 $O/$tag.err; }
EXTRA=""; line rccl1 SEQWIN_BENCH_FORCE_DIST=1 SEQWIN_DIST_FORCE_COLLECTIVES=1; line rccl1_nolimit SEQWIN_BENCH_FORCE_DIST=1 SEQWIN_DIST_FORCE_COLLECTIVES=1 SEQWIN_DIST_MSG_LIMIT_MB=100000
