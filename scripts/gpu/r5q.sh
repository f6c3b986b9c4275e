# round 5, call Q: what the chained look-back costs k_nodes (timing ablation: -DSW_NODES_NO_LOOKBACK gives wrong node numbers)
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5q; mkdir -p $O; cd /tmp
for v in shipped nolook shipped2 nolook2; do
  lib=$R/seqwin_amd/libseqwin_hip.so; case $v in nolook*) lib=$R/ab_live/libseqwin_hip_nolook.so ;; esac
  SEQWIN_AMD_LIB=$lib timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$v -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/prof_$v.json 2> $O/prof_$v.err
  echo "prof $v rc=$?"
  f=$(find $O/prof_$v -name "*kernel_stats.csv" | head -n 1); [ -n "$f" ] && grep "k_nodes" $f | cut -d, -f1-4 | cut -c1-60,200-260
  rm -rf $O/prof_$v
done
