# round 3, call h: at which size does the RCCL world-1 run of the sharded path go wrong?
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3h; mkdir -p $O; cd $R
line() { tag=$1; shift; env "$@" timeout -k 10 300 python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline $EXTRA 2>$O/$tag.err | grep "^{" > $O/$tag.json; python3 -c "
import json; d=json.load(open('$O/$tag.json')); print('$tag', d['ms_per_step'], d['counts'], d['checksums'], {k: round(v,1) for k,v in d['stages_ms'].items() if 'wall' in k})" || tail -n 5 $O/$tag.err; }
for g in 5000 9000 11000 13000; do EXTRA="--genomes $g"; line direct_$g A=1; line rccl_$g SEQWIN_BENCH_FORCE_DIST=1 SEQWIN_DIST_FORCE_COLLECTIVES=1; done
