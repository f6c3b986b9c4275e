# round 5, call AE: the round's profile set again on the final library (kernel stats + HBM traffic of the default workload, of one GPU's share of
# random100k at k = 15 / 19 / 31 and of salmonella500 at w = 10; SQ counters of the sketch kernel) and the bench lines that go to profiles/
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5ae; mkdir -p $O; cd /tmp
prof() {   # prof <set> <kind> <bench args...>
  set_=$1; kind=$2; shift 2
  case $kind in
    stats) opts="--kernel-trace --stats"; st="--steps 4 --warmup 1" ;;
    pmc)   opts="--pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY"; st="--steps 1 --warmup 1" ;;
    fetch) opts="--pmc FETCH_SIZE"; st="--steps 1 --warmup 1" ;;
    write) opts="--pmc WRITE_SIZE"; st="--steps 1 --warmup 1" ;;
  esac
  timeout -k 10 240 rocprofv3 $opts --output-format csv -d $O/$set_/$kind -- python3 $R/bench.py $st --no-cpu-baseline "$@" > $O/${set_}_$kind.log 2>&1
  rc=$?; echo "$set_ $kind rc=$rc"
  find $O/$set_/$kind -type f ! -name "*kernel_stats.csv" ! -name "*counter_collection.csv" -delete 2>/dev/null
  return $rc
}
for kind in stats pmc fetch write; do prof bacteria15k $kind || exit 1; done
for k in 19 15 31; do for kind in stats fetch write; do prof random100k_k$k $kind --workload random100k -k $k || exit 1; done; done
for kind in stats fetch write; do prof salmonella500_w10 $kind --workload salmonella500 -w 10 || exit 1; done
du -sh $O
cd $R
for k in 15 19 31; do
  timeout -k 10 200 python3 bench.py --workload random100k -k $k --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_random100k_k$k.json 2> $O/bench_random100k_k$k.err || { echo "bench k$k failed"; tail $O/bench_random100k_k$k.err; exit 1; }
  python3 -c "
import json; d=json.load(open('$O/bench_random100k_k$k.json')); print('k$k', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('traffic_source'), d['parity']['n1_checksums_equal'], d['parity']['full_size_vs'])"
done
timeout -k 10 200 python3 bench.py --workload salmonella500 -w 10 --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_salmonella500_w10.json 2> $O/bench_salmonella500_w10.err; echo "bench sal w10 rc=$?"
timeout -k 10 400 python3 bench.py --steps 10 --warmup 2 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; python3 -c "
import json; d=json.load(open('$O/bench_default.json')); print(d['value'], d['ms_per_step'], d['roofline'], d['cpu_baseline']['value'], d['e2e']['value'], d['e2e']['first_call'], d['e2e']['by_n_cpu'], d['e2e']['gz']['routes']['host'], d['parity']['equal'])"
