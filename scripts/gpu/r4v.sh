# round 4, call V: direct scatter of the ranks (default up to 2^25 occurrences) against the bucketed unsort + keys from the buckets, at 6 / 12 / 24 M occurrences
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4v; mkdir -p $O; cd $R
for g in 128 256 512; do
for v in direct bucketed direct bucketed; do
  unset SEQWIN_AMD_UNSORT_DIRECT
  if [ $v = bucketed ]; then export SEQWIN_AMD_UNSORT_DIRECT=20; fi
  timeout -k 10 300 python3 bench.py --workload salmonella500 --genomes $g --steps 40 --warmup 5 --no-cpu-baseline > $O/bench_$v.json 2> $O/bench_$v.err || exit 1
  python3 -c "import json; d=json.loads([l for l in open('$O/bench_$v.json') if l.startswith('{')][-1]); print($g, '$v', d['counts']['kmers'], d['ms_per_step'], d['stages_ms']['nodes_ms'], d['stages_ms']['edges_ms'])"
done
done
