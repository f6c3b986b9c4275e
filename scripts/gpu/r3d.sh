# round 3, call d: full GPU suite on the reworked multi-GPU API + sketch latency experiments (DPP scan, sleep, phase stamps)
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3d; mkdir -p $O; cd $R
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -n 4 $O/tests.log
run() { tag=$1; lib=$2; shift 2; env SEQWIN_AMD_LIB=$lib "$@" timeout -k 10 300 python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/$tag.json 2>$O/$tag.err; python3 -c "
import json; d=json.load(open('$O/$tag.json')); print('$tag', d['value'], d['ms_per_step'], d['stages_ms'], d.get('parity'))"; grep stamps $O/$tag.err | tail -n 2; }
run new $R/seqwin_amd/libseqwin_hip.so A=1 && run nodpp $R/ab/libseqwin_hip_nodpp.so A=1 && run sleep16 $R/ab/libseqwin_hip_sleep16.so A=1 && run stamps $R/ab/libseqwin_hip_stamps.so SEQWIN_AMD_STAMPS=1 && run new2 $R/seqwin_amd/libseqwin_hip.so A=1
