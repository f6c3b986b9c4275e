# round 3, call f: device-side plan + one-round-trip tile start + L2 warming; full GPU suite, fuzz (gaps!), A/B
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3f; mkdir -p $O; cd $R
timeout -k 10 900 python3 -m pytest tests -m gpu -q -x > $O/tests.log 2>&1; echo "tests rc=$?"; tail -n 6 $O/tests.log
timeout -k 10 100 python3 tests/tools/fuzz_gpu.py 60 51 > $O/fuzz_default.log 2>&1 &
SEQWIN_AMD_RC=3 SEQWIN_AMD_SLOT_CAP=3 timeout -k 10 100 python3 tests/tools/fuzz_gpu.py 60 52 > $O/fuzz_rc3.log 2>&1 &
FUZZ_DIST=1 timeout -k 10 100 python3 tests/tools/fuzz_gpu.py 60 53 > $O/fuzz_dist.log 2>&1 &
SEQWIN_AMD_SKETCH=tails timeout -k 10 100 python3 tests/tools/fuzz_gpu.py 60 54 > $O/fuzz_tails.log 2>&1 &
wait; tail -n 1 $O/fuzz_default.log $O/fuzz_rc3.log $O/fuzz_dist.log $O/fuzz_tails.log
run() { tag=$1; lib=$2; shift 2; env SEQWIN_AMD_LIB=$lib "$@" timeout -k 10 300 python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/$tag.json 2>$O/$tag.err; python3 -c "
import json; d=json.load(open('$O/$tag.json')); print('$tag', d['value'], d['ms_per_step'], d['stages_ms'], 'plan_ms', d['plan_ms'], d.get('parity'))"; grep stamps $O/$tag.err | tail -n 1; }
run new $R/seqwin_amd/libseqwin_hip.so A=1 && run nopf $R/ab/libseqwin_hip_nopf.so A=1 && run pf1024 $R/ab/libseqwin_hip_pf1024.so A=1 && run pf4096 $R/ab/libseqwin_hip_pf4096.so A=1 && run stamps $R/ab/libseqwin_hip_stamps.so SEQWIN_AMD_STAMPS=1 && run new2 $R/seqwin_amd/libseqwin_hip.so A=1
