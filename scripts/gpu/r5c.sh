# round 5, call C: cost of the always-on order guard (A/B: -DSW_NO_ORDER_GUARD), the default bench line with its new legs
# (all-core CPU baseline, e2e over n_cpu, .gz leg), one GPU's share of random100k at k = 19 pinned to the compiled reference,
# and the random100k lines for k = 15 / 19 / 31 with the thread-per-descent repair
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5c; mkdir -p $O; cd $R
{ cat /sys/fs/cgroup/cpu.max; cat /sys/fs/cgroup/cpu.stat | head -8; nproc; } > $O/cpu.txt 2>&1
run() { tag=$1; lib=$2; shift 2; SEQWIN_AMD_LIB=$lib timeout -k 10 300 python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline "$@" > $O/$tag.json 2>$O/$tag.err; python3 -c "
import json; d=json.load(open('$O/$tag.json')); print('$tag', d['value'], d['ms_per_step'], d['stages_ms'], d.get('parity'))"; }
run guard_a $R/seqwin_amd/libseqwin_hip.so && run noguard_a $R/ab_live/libseqwin_hip_noguard.so && run guard_b $R/seqwin_amd/libseqwin_hip.so && run noguard_b $R/ab_live/libseqwin_hip_noguard.so
for k in 19 15 31; do run r100k_k$k $R/seqwin_amd/libseqwin_hip.so --workload random100k -k $k --steps 4; done
timeout -k 10 300 python3 bench.py --steps 10 --warmup 2 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; python3 -c "
import json; d=json.load(open('$O/bench_default.json')); print(d['value'], d['ms_per_step'], d['cpu_baseline'], d['e2e'], d['parity'])"
timeout -k 10 500 python3 scripts/pin_fullsize_ref.py --workload random100k -k 19 -w 200 --size-from gpurun_out/r5b/pin_random100k_k19_probe.json --out $O/pin_random100k_k19.json > $O/pin_random100k_k19.log 2>&1
echo "pin r100k rc=$?"; tail -n 6 $O/pin_random100k_k19.log
