# round 6, call E: ragged500 (tests, pin to the compiled reference, bench line + kernel stats), the placement probe, the sparse-stage prototype
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6e; mkdir -p $O; cd $R
timeout -k 10 200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "ragged" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -n 3 $O/tests.log
bash scripts/gpu/pins.sh r6e ragged500:21:200 || exit 1
timeout -k 10 200 python3 bench.py --workload ragged500 > $O/bench_ragged500.json 2> $O/bench_ragged500.err; echo "bench ragged rc=$?"
timeout -k 10 120 python3 bench.py --workload salmonella500 --no-cpu-baseline > $O/bench_salmonella500.json 2> $O/bench_salmonella500.err; echo "bench salmonella rc=$?"
python3 -c "
import json
for n in ('ragged500','salmonella500'):
    d=json.load(open('$O/bench_%s.json'%n)); print(n, d['value'], d['ms_per_step'], d['stages_ms'], d.get('tiles'), d.get('parity'), d['config']['workload'][:90])"
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_ragged/stats -- python3 $R/bench.py --workload ragged500 --steps 4 --warmup 1 --no-cpu-baseline > $O/prof_ragged.log 2>&1; echo "prof rc=$?"; cd $R
timeout -k 10 200 python3 tests/tools/placement_probe.py 6 3 > $O/placement_probe.log 2>&1; echo "probe rc=$?"; cat $O/placement_probe.log | tail -n 8
cd scripts/micro && ./sparse_stage 200000 200 16 > $O/sparse_d16.log 2>&1; echo "sparse16 rc=$?"; cat $O/sparse_d16.log
./sparse_stage 200000 200 32 > $O/sparse_d32.log 2>&1; cat $O/sparse_d32.log
cd /tmp && rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/prof_sparse -- $R/scripts/micro/sparse_stage 200000 200 16 > $O/prof_sparse.log 2>&1; echo "prof sparse rc=$?"
