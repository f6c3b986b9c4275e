# round 4, call Y: one GPU's share of the 15k job at P = 8 (1 875 genomes): sharded form at N = 1 without and with every collective issued over RCCL
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4y; mkdir -p $O; cd $R
export SEQWIN_BENCH_FORCE_DIST=1
for v in nocoll coll nocoll coll; do
  if [ $v = coll ]; then export SEQWIN_DIST_FORCE_COLLECTIVES=1; else unset SEQWIN_DIST_FORCE_COLLECTIVES; fi
  timeout -k 10 300 python3 bench.py --genomes 1875 --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_$v.json 2> $O/bench_$v.err || exit 1
  python3 -c "import json; d=json.loads([l for l in open('$O/bench_$v.json') if l.startswith('{')][-1]); print('$v', d['value'], d['ms_per_step'], d['dist']['phases_ms_max_over_ranks'])"
done
