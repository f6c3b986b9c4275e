export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3m; mkdir -p $O; cd $R
{ python3 scripts/dbg/sort_time.py 745 54
SEQWIN_AMD_RADIX_BITS=8 python3 scripts/dbg/sort_time.py 745 54
SEQWIN_AMD_RADIX_SHAPE=256x8 python3 scripts/dbg/sort_time.py 745 54
SEQWIN_AMD_RADIX_SHAPE=1024x8 python3 scripts/dbg/sort_time.py 745 54
python3 scripts/dbg/sort_time.py 745 16
SEQWIN_AMD_RADIX_SHAPE=256x8 python3 scripts/dbg/sort_time.py 745 16
SEQWIN_AMD_SORT=rocprim python3 scripts/dbg/sort_time.py 745 54
SEQWIN_AMD_RADIX_SHAPE=256x8 timeout -k 10 300 python3 -m pytest tests -m gpu -x -q -k "sort_keys64" 2>&1 | tail -n 2; } 2>&1 | grep -v amdgpu.ids | tee $O/sort_shapes.log
