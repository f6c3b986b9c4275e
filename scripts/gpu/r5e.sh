# round 5, call E: the GPU suite on the library with dense node parts / k_finish_nodes, staged multi-device routes and the A/B
# losers compiled out; per-kernel times with and without the order guards; one GPU's share of random100k (k = 19) pinned to the
# compiled reference (the call before lost its outputs: gpurun_out/ was above the 64 MiB merge limit)
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5e; mkdir -p $O; cd $R
timeout -k 10 420 python3 -m pytest tests -q -m gpu -x > $O/tests.log 2>&1; echo "suite rc=$?"; tail -n 6 $O/tests.log
cd /tmp
for v in guard noguard; do
  lib=$R/seqwin_amd/libseqwin_hip.so; [ $v = noguard ] && lib=$R/ab_live/libseqwin_hip_noguard.so
  SEQWIN_AMD_LIB=$lib timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$v -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline > $O/prof_$v.json 2> $O/prof_$v.err
  echo "prof $v rc=$?"
  f=$(find $O/prof_$v -name "*kernel_stats.csv" | head -n 1); [ -n "$f" ] && cp $f $O/kernel_stats_$v.csv && head -n 16 $f | cut -d, -f1-4 | cut -c1-120
  rm -rf $O/prof_$v
done
cd $R
for k in 19; do SEQWIN_AMD_LIB=$R/seqwin_amd/libseqwin_hip.so timeout -k 10 300 python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --workload random100k -k $k > $O/r100k_k$k.json 2>$O/r100k_k$k.err; python3 -c "
import json; d=json.load(open('$O/r100k_k$k.json')); print('r100k k$k', d['value'], d['ms_per_step'], d['stages_ms'])"; done
timeout -k 10 500 python3 scripts/pin_fullsize_ref.py --workload random100k -k 19 -w 200 --size-from "tests/golden/bench_checksums_ref.json#random100k/k19/w200@2500" --out $O/pin_random100k_k19.json > $O/pin_random100k_k19.log 2>&1
echo "pin r100k rc=$?"; tail -n 5 $O/pin_random100k_k19.log
