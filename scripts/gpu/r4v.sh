# round 4, call V: k_pen_bits launched behind the adjacency kernel (default) against right after the nodes stage (SEQWIN_AMD_COUNTS_EARLY=1)
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4v; mkdir -p $O; cd $R
for w in "--workload random100k -k 19" ""; do
for v in early behind early behind; do
  unset SEQWIN_AMD_COUNTS_EARLY
  if [ $v = early ]; then export SEQWIN_AMD_COUNTS_EARLY=1; fi
  timeout -k 10 300 python3 bench.py $w --steps 6 --warmup 2 --no-cpu-baseline > $O/bench_$v.json 2> $O/bench_$v.err || exit 1
  python3 -c "import json; d=json.loads([l for l in open('$O/bench_$v.json') if l.startswith('{')][-1]); print('$w', '$v', d['value'], d['ms_per_step'], d['stages_ms'], d.get('parity'))"
done
done
