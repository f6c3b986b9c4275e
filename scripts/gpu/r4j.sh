# round 4, call J: look-back schemes side by side on one box (A: 8-byte records, 4 per step, first step requested early; C: 4-byte aggregates, 8 per step + one inclusive probe)
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4j; mkdir -p $O; cd $R
T="timeout -k 10 200 python3 tests/tools/sort_time.py 745 45"
for i in 1 2 3; do for v in lbA lbC; do SEQWIN_AMD_LIB=$R/ab/libseqwin_hip_$v.so $T > $O/st_$v.log 2>&1; echo "$v $(grep bits= $O/st_$v.log | sed 's/.*bits=45: //')"; done; done
