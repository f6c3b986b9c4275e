# round 4, call E: radix pass experiments (pipelined atomics; shapes; look-back width) and the bench line with the new e2e leg
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4e; mkdir -p $O; cd $R
T="timeout -k 10 200 python3 tests/tools/sort_time.py 745 45"
$T > $O/st_default.log 2>&1; grep bits= $O/st_default.log
SEQWIN_AMD_RADIX_SHAPE=512x9 $T > $O/st_512x9.log 2>&1; grep bits= $O/st_512x9.log
SEQWIN_AMD_RADIX_SHAPE=1024x8 $T > $O/st_1024x8.log 2>&1; grep bits= $O/st_1024x8.log
SEQWIN_AMD_RADIX_BITS=8 $T > $O/st_512x8.log 2>&1; grep bits= $O/st_512x8.log
SEQWIN_AMD_LIB=$R/ab/libseqwin_hip_look8.so $T > $O/st_look8.log 2>&1; echo look8; grep bits= $O/st_look8.log
SEQWIN_AMD_LIB=$R/ab/libseqwin_hip_look2.so $T > $O/st_look2.log 2>&1; echo look2; grep bits= $O/st_look2.log
SEQWIN_AMD_STAMPS=1 SEQWIN_AMD_LIB=$R/ab/libseqwin_hip_rsst.so $T > $O/stamps.log 2>&1; grep "rs stamps" $O/stamps.log | tail -n 2
SEQWIN_AMD_RADIX_SHAPE=512x9 SEQWIN_AMD_STAMPS=1 SEQWIN_AMD_LIB=$R/ab/libseqwin_hip_rsst.so $T > $O/stamps512.log 2>&1; grep "rs stamps" $O/stamps512.log | tail -n 2
timeout -k 10 600 python3 bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -n 3 $O/bench.err
python3 -c "
import json; d=json.load(open('$O/bench.json')); print(d['value'], d['ms_per_step'], d['stages_ms'], d.get('parity')); print(d['cpu_baseline']); print(d['e2e'])"
