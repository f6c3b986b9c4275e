# round 5, call L: order guards with scalar accumulation -- the fault-injection tests, then per-kernel times with and without
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5l; mkdir -p $O; cd $R
timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "order_guard or bench_line" > $O/guard.log 2>&1; echo "guard rc=$?"; tail -n 3 $O/guard.log
cd /tmp
for v in guard noguard guard2 noguard2; do
  lib=$R/seqwin_amd/libseqwin_hip.so; case $v in noguard*) lib=$R/ab_live/libseqwin_hip_noguard.so ;; esac
  SEQWIN_AMD_LIB=$lib timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$v -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline > $O/prof_$v.json 2> $O/prof_$v.err
  echo "prof $v rc=$?"
  f=$(find $O/prof_$v -name "*kernel_stats.csv" | head -n 1); [ -n "$f" ] && cp $f $O/kernel_stats_$v.csv
  rm -rf $O/prof_$v
done
