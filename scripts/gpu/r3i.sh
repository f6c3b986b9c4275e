export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3i; mkdir -p $O; cd $R
SEQWIN_DIST_DEBUG=1 SEQWIN_BENCH_FORCE_DIST=1 SEQWIN_DIST_FORCE_COLLECTIVES=1 timeout -k 10 300 python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --genomes 3000 > $O/dbg.log 2>&1; grep "dist debug\|counts" $O/dbg.log | cut -c1-600
