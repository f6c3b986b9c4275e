# round 5, call Z: is the e2e ingest bound by the parsers or by the thread that copies packed words into the pinned ring?
# (SEQWIN_AMD_DEBUG_TIMING prints how long the sink thread waited for the parsers)
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5z; mkdir -p $O; cd $R
for nc in 16 32 64; do
  SEQWIN_AMD_DEBUG_TIMING=1 timeout -k 10 300 python3 tests/tools/e2e_ingest_ab.py 2048 $nc > $O/ab_$nc.txt 2>&1; echo "ab $nc rc=$?"; grep -E "read\(\)|sink thread|ingest_to_device" $O/ab_$nc.txt | head -40
done
