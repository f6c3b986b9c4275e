# round 4, call V: where radix.hip's sorts (+ stage order) start to pay: genome counts of the 512-genome workload; default (pairs from 2^20, keys from 2^23) against all-rocPRIM and all-own
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4v; mkdir -p $O; cd $R
for g in 8 16 32 64 128 256 512; do
for v in rocprim default own; do
  unset SEQWIN_AMD_SORT
  if [ $v != default ]; then export SEQWIN_AMD_SORT=$v; fi
  timeout -k 10 300 python3 bench.py --workload salmonella500 --genomes $g --steps 40 --warmup 5 --no-cpu-baseline > $O/bench_$v.json 2> $O/bench_$v.err || exit 1
  python3 -c "import json; d=json.loads([l for l in open('$O/bench_$v.json') if l.startswith('{')][-1]); print($g, '$v', d['counts']['kmers'], d['ms_per_step'], d['stages_ms']['nodes_ms'], d['stages_ms']['edges_ms'], d['stages_ms']['order_ms'])"
done
done
