R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2s; mkdir -p $O; cd $R
timeout -k 10 300 python -m pytest tests -m gpu -x -q -k "sort_keys64 or large_config or two_phase" > $O/pytest.log 2>&1 || { tail -n 30 $O/pytest.log; exit 1; }
tail -n 3 $O/pytest.log
python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/bench_persistent.json 2> $O/bench_persistent.err && python3 -c "
import json,sys; d=json.load(open('$O/bench_persistent.json')); print('persistent', d['value'], d['ms_per_step'], d['stages_ms'], d['parity'])"
SEQWIN_AMD_RADIX_KERNEL=classic python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/bench_classic.json 2> $O/bench_classic.err && python3 -c "
import json,sys; d=json.load(open('$O/bench_classic.json')); print('classic   ', d['value'], d['ms_per_step'], d['stages_ms'], d['parity'])"
