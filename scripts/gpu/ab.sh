# A/B timing of library builds on the default workload.  Variants are built with tests/tools/build_variant.sh <name> <flags>
# (ab_live/libseqwin_hip_<name>.so, shipped to the GPU box).  usage (through gpurun): bash scripts/gpu/ab.sh <tag> <name> [<name> ...]
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O; cd $R; shift
run() { tag=$1; lib=$2; SEQWIN_AMD_STAMPS=1 SEQWIN_AMD_LIB=$lib timeout -k 10 300 python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/$tag.json 2>$O/$tag.err; python3 -c "
import json; d=json.load(open('$O/$tag.json')); print('$tag', d['value'], d['ms_per_step'], d['stages_ms'], 'plan_ms', d['plan_ms'], d.get('parity'))"; grep stamps $O/$tag.err | tail -n 1; }
run shipped $R/seqwin_amd/libseqwin_hip.so
for v in "$@"; do run $v $R/ab_live/libseqwin_hip_$v.so; done
run shipped2 $R/seqwin_amd/libseqwin_hip.so
