# round 5, call H: the GPU suite, then differential fuzz campaigns on the round's library (dense node parts, thread repair, order
# guards, adaptive second skipped digit of the edge sort, host inflate): default set and the multi-device / stage sets
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5h; mkdir -p $O; cd $R
timeout -k 10 420 python3 -m pytest tests -q -m gpu -x > $O/tests.log 2>&1; echo "suite rc=$?"; tail -n 4 $O/tests.log
bash scripts/gpu/fuzz.sh r5h/fuzz_default 200 && bash scripts/gpu/fuzz.sh r5h/fuzz_multi 150 multi && bash scripts/gpu/fuzz.sh r5h/fuzz_stage 150 stage
for k in 19; do timeout -k 10 300 python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --workload random100k -k $k > $O/r100k_k$k.json 2>$O/r100k_k$k.err; python3 -c "
import json; d=json.load(open('$O/r100k_k$k.json')); print('r100k k$k', d['value'], d['ms_per_step'], d['stages_ms'], d['parity'])"; done
timeout -k 10 300 python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --workload salmonella500 -w 10 > $O/w10.json 2>$O/w10.err; python3 -c "
import json; d=json.load(open('$O/w10.json')); print('w10', d['value'], d['ms_per_step'], d['stages_ms'])"
