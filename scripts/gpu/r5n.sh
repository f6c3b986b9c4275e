# round 5, call N: A/B of three small variants on the default set and on the random100k share: endpoint hashes of the edges by
# non-temporal loads; k_nodes tiles of 3 / 6 rows (4 is the default)
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5n; mkdir -p $O; cd $R
run() { tag=$1; lib=$2; shift 2; SEQWIN_AMD_LIB=$lib timeout -k 10 300 python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline "$@" > $O/$tag.json 2>$O/$tag.err; python3 -c "
import json; d=json.load(open('$O/$tag.json')); print('$tag', d['value'], d['ms_per_step'], d['stages_ms'], (d.get('parity') or {}).get('n1_checksums_equal'))"; }
S=$R/seqwin_amd/libseqwin_hip.so
for rep in a b; do
  run shipped_$rep $S
  for v in edgent rows3 rows6; do run ${v}_$rep $R/ab_live/libseqwin_hip_$v.so; done
done
for rep in a b; do
  run r100k_shipped_$rep $S --workload random100k -k 19 --steps 4
  for v in edgent rows3 rows6; do run r100k_${v}_$rep $R/ab_live/libseqwin_hip_$v.so --workload random100k -k 19 --steps 4; done
done
