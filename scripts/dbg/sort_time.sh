cd $GRAFT_REPO_ROOT   # (the RADIX_DEBUG lines need a library built with -DSW_RADIX_ABLATION)
python3 scripts/dbg/sort_time.py
SEQWIN_AMD_RADIX_KERNEL=classic python3 scripts/dbg/sort_time.py
SEQWIN_AMD_RADIX_DEBUG=1 python3 scripts/dbg/sort_time.py
python3 scripts/dbg/sort_time.py 745 16
SEQWIN_AMD_RADIX_KERNEL=classic python3 scripts/dbg/sort_time.py 745 16
timeout -k 10 300 python -m pytest tests -m gpu -x -q -k "sort_keys64" 2>&1 | tail -n 2
