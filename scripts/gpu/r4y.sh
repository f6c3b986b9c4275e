# round 4, call Y: one GPU's share of the 15k job at P = 8 (1 875 genomes): direct build and the sharded form at N = 1
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4y; mkdir -p $O; cd $R
for v in direct dist direct dist; do
  if [ $v = dist ]; then export SEQWIN_BENCH_FORCE_DIST=1; else unset SEQWIN_BENCH_FORCE_DIST; fi
  timeout -k 10 300 python3 bench.py --genomes 1875 --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_$v.json 2> $O/bench_$v.err || exit 1
  python3 -c "import json; d=json.loads([l for l in open('$O/bench_$v.json') if l.startswith('{')][-1]); print('$v', d['value'], d['ms_per_step'], d['stages_ms'] if '$v'=='direct' else d['dist']['phases_ms_max_over_ranks'])"
done
