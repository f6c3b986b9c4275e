export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3n; mkdir -p $O; cd $R
{ SEQWIN_AMD_LIB=$R/ab/libseqwin_hip_rsstamps.so SEQWIN_AMD_STAMPS=1 python3 scripts/dbg/sort_time.py 745 54
SEQWIN_AMD_LIB=$R/ab/libseqwin_hip_rsstamps.so SEQWIN_AMD_STAMPS=1 python3 scripts/dbg/sort_time.py 745 16; } 2>&1 | grep -v amdgpu.ids | tee $O/rs_stamps.log | tail -n 12
