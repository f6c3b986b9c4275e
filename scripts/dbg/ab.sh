R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2h; mkdir -p $O; cd $R
run() { tag=$1; shift; env "$@" python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline $EXTRA > $O/$tag.json 2>$O/$tag.err; python3 -c "
import json,sys; d=json.load(open('$O/$tag.json')); print('$tag', d['value'], d['ms_per_step'], d['stages_ms'], d.get('parity'))"; }
EXTRA=""; run base15k A=1; run table15k SEQWIN_AMD_RANKS=table; run direct15k SEQWIN_AMD_UNSORT_DIRECT=34
EXTRA="--workload salmonella500 --steps 20 --warmup 2"; run base_salm A=1; run table_salm SEQWIN_AMD_RANKS=table; run direct_salm SEQWIN_AMD_UNSORT_DIRECT=34
