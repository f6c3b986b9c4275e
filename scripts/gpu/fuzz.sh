# differential fuzz campaign on the GPU box: five processes share the card (six make four of them crawl).
# usage (through gpurun): bash scripts/gpu/fuzz.sh <tag> [seconds] [gz|multi|stage]      gz: the two device-gzip-ingest campaigns instead;
#                                                                                 multi: sw_build over logical devices (SEQWIN_DEVICES) instead; stage: the node sort reading the sketch stage (SEQWIN_AMD_ORDER=stage)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O; cd $R; T=${2:-240}
export SEQWIN_AMD_LIB=${SEQWIN_AMD_LIB:-$R/seqwin_amd/libseqwin_hip_test.so}   # (the knobs of the campaigns are test hooks: the test library)
if [ "$3" = "gz" ]; then
FUZZ_GZ=1 SEQWIN_AMD_DEVICE_INFLATE=1 python3 tests/tools/fuzz_gpu.py $T 26 > $O/fuzz_device_gz.log 2>&1 &
FUZZ_GZ=1 SEQWIN_AMD_DEVICE_INFLATE=1 SEQWIN_AMD_RC=3 SEQWIN_AMD_SLOT_CAP=3 python3 tests/tools/fuzz_gpu.py $T 27 > $O/fuzz_device_gz_rc3.log 2>&1 &
elif [ "$3" = "multi2" ]; then   # r06 soak of the multi-device path: 2 and 4 logical devices, with POOL_DEBUG=$POOL_DEBUG (export it: 0 / 1)
SEQWIN_AMD_POOL_DEBUG=${POOL_DEBUG:-0} SEQWIN_DEVICES=0,0 python3 tests/tools/fuzz_gpu.py $T 51 > $O/fuzz_devices2_a.log 2>&1 &
SEQWIN_AMD_POOL_DEBUG=${POOL_DEBUG:-0} SEQWIN_DEVICES=0,0 python3 tests/tools/fuzz_gpu.py $T 52 > $O/fuzz_devices2_b.log 2>&1 &
SEQWIN_AMD_POOL_DEBUG=${POOL_DEBUG:-0} SEQWIN_DEVICES=0,0,0,0 python3 tests/tools/fuzz_gpu.py $T 53 > $O/fuzz_devices4_a.log 2>&1 &
SEQWIN_AMD_POOL_DEBUG=${POOL_DEBUG:-0} SEQWIN_DEVICES=0,0,0,0 SEQWIN_DIST_HASH_ROUTE=requests python3 tests/tools/fuzz_gpu.py $T 54 > $O/fuzz_devices4_requests.log 2>&1 &
elif [ "$3" = "multi" ]; then
SEQWIN_DEVICES=0,0 python3 tests/tools/fuzz_gpu.py $T 31 > $O/fuzz_devices2.log 2>&1 &
SEQWIN_DEVICES=0,0,0 SEQWIN_DIST_HASH_ROUTE=requests SEQWIN_AMD_SORT=own python3 tests/tools/fuzz_gpu.py $T 32 > $O/fuzz_devices3_requests_own.log 2>&1 &
SEQWIN_DEVICES=0,0,0,0,0 SEQWIN_AMD_RC=3 SEQWIN_AMD_SLOT_CAP=3 python3 tests/tools/fuzz_gpu.py $T 33 > $O/fuzz_devices5_rc3.log 2>&1 &
SEQWIN_AMD_SORT=own SEQWIN_AMD_RADIX_RANK=ballot python3 tests/tools/fuzz_gpu.py $T 34 > $O/fuzz_own_ballot.log 2>&1 &
elif [ "$3" = "stage" ]; then
SEQWIN_AMD_ORDER=stage python3 tests/tools/fuzz_gpu.py $T 41 > $O/fuzz_stage.log 2>&1 &
SEQWIN_AMD_ORDER=stage SEQWIN_AMD_RC=3 SEQWIN_AMD_SLOT_CAP=3 python3 tests/tools/fuzz_gpu.py $T 42 > $O/fuzz_stage_rc3.log 2>&1 &
SEQWIN_AMD_ORDER=stage SEQWIN_AMD_SORT=own SEQWIN_AMD_UNSORT_DIRECT=4 SEQWIN_AMD_WINDOW_SPLIT=8,4 python3 tests/tools/fuzz_gpu.py $T 43 > $O/fuzz_stage_own.log 2>&1 &
SEQWIN_AMD_ORDER=stage SEQWIN_AMD_SKETCH=nosmall SEQWIN_AMD_NO_PACKED_EDGES=1 SEQWIN_AMD_CHECK_ORDER=1 python3 tests/tools/fuzz_gpu.py $T 44 > $O/fuzz_stage_knobs.log 2>&1 &
else
python3 tests/tools/fuzz_gpu.py $T 21 > $O/fuzz_default.log 2>&1 &
SEQWIN_AMD_SORT=own SEQWIN_AMD_EDGE_SKIP_PASSES=2 SEQWIN_AMD_UNSORT_DIRECT=4 SEQWIN_AMD_WINDOW_SPLIT=8,4 python3 tests/tools/fuzz_gpu.py $T 22 > $O/fuzz_unsort_winsplit.log 2>&1 &
SEQWIN_AMD_UNSORT_DIRECT=4 SEQWIN_AMD_SORT_KEYBITS=10 SEQWIN_AMD_NO_PACKED_EDGES=1 SEQWIN_AMD_CHECK_ORDER=1 python3 tests/tools/fuzz_gpu.py $T 23 > $O/fuzz_knobs.log 2>&1 &
FUZZ_DIST=1 SEQWIN_AMD_SORT=own SEQWIN_AMD_RADIX_BITS=9 SEQWIN_AMD_UNSORT_DIRECT=6 python3 tests/tools/fuzz_gpu.py $T 24 > $O/fuzz_dist.log 2>&1 &
FUZZ_LOWMEM=1 SEQWIN_AMD_LOWMEM_CHUNK_MBP=0 SEQWIN_AMD_RANKS=table SEQWIN_AMD_RC=3 python3 tests/tools/fuzz_gpu.py $T 25 > $O/fuzz_lowmem_table_rc3.log 2>&1 &
fi
for i in $(seq 1 40); do sleep 30; echo "t=$((i*30))s"; kill -0 $! 2>/dev/null || break; done
wait
tail -n 2 $O/fuzz_*.log
