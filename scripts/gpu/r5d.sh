# round 5, call D: one GPU's share of random100k (12 500 iid genomes, k = 19) pinned to the compiled reference; per-kernel times of the
# build with and without the order guards (rocprofv3 --kernel-trace --stats); the multi-device tests (routes, staged fallback)
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5d; mkdir -p $O; cd $R
timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "multi_device or order_guard" > $O/multi.log 2>&1; echo "multi rc=$?"; tail -n 4 $O/multi.log
cd /tmp
for v in guard noguard; do
  lib=$R/seqwin_amd/libseqwin_hip.so; [ $v = noguard ] && lib=$R/ab_live/libseqwin_hip_noguard.so
  SEQWIN_AMD_LIB=$lib timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof_$v -o $v -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline > $O/prof_$v.json 2> $O/prof_$v.err
  echo "prof $v rc=$?"
  f=$(find $O/prof_$v -name "*kernel_stats.csv" | head -n 1); [ -n "$f" ] && head -n 14 $f | cut -d, -f1-4 | cut -c1-110
done
cd $R
timeout -k 10 700 python3 scripts/pin_fullsize_ref.py --workload random100k -k 19 -w 200 --size-from "tests/golden/bench_checksums_ref.json#random100k/k19/w200@2500" --out $O/pin_random100k_k19.json > $O/pin_random100k_k19.log 2>&1
echo "pin r100k rc=$?"; tail -n 8 $O/pin_random100k_k19.log
