# round 3, call a: parity of the reworked sketch kernel + A/B timing of its three changes on bacteria15k
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3b; mkdir -p $O; cd $R
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -n 3 $O/tests.log
timeout -k 10 100 python3 tests/tools/fuzz_gpu.py 70 31 > $O/fuzz_default.log 2>&1 &
SEQWIN_AMD_RC=3 SEQWIN_AMD_SLOT_CAP=3 timeout -k 10 100 python3 tests/tools/fuzz_gpu.py 70 32 > $O/fuzz_rc3.log 2>&1 &
wait; tail -n 2 $O/fuzz_default.log $O/fuzz_rc3.log
run() { tag=$1; lib=$2; SEQWIN_AMD_LIB=$lib timeout -k 10 300 python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/$tag.json 2>$O/$tag.err; python3 -c "
import json; d=json.load(open('$O/$tag.json')); print('$tag', d['value'], d['ms_per_step'], d['stages_ms'], d.get('parity'))"; }
run new $R/seqwin_amd/libseqwin_hip.so && run bonly $R/ab/libseqwin_hip_bonly.so && run old $R/ab/libseqwin_hip_old.so && run nohoist $R/ab/libseqwin_hip_nohoist.so && run hoistonly $R/ab/libseqwin_hip_hoistonly.so && run new2 $R/seqwin_amd/libseqwin_hip.so
