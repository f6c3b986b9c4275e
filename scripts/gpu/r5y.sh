# round 5, call Y: k_unsort_adj places a bucket's candidates with one global atomic per workgroup and bucket (two-phase, no staging area):
# GPU suite, then the bench lines of random100k at k = 15 / 19 / 31 and of the default workload (checksums against the reference's),
# the k = 15 profile set again, and a fuzz set (its generator has low-complexity genomes: many candidates per bucket, staging overflow)
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5y; mkdir -p $O; cd $R
timeout -k 10 480 python3 -m pytest tests -q -m gpu -x > $O/tests.log 2>&1; rc=$?; echo "suite rc=$rc"; tail -n 4 $O/tests.log; [ $rc = 0 ] || exit 1
for k in 15 19 31; do
  timeout -k 10 200 python3 bench.py --workload random100k -k $k --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_r100k_k$k.json 2> $O/bench_r100k_k$k.err || { echo "bench k$k failed"; tail $O/bench_r100k_k$k.err; exit 1; }
  python3 -c "
import json; d=json.load(open('$O/bench_r100k_k$k.json')); print('k$k', d['value'], d['ms_per_step'], d.get('parity'))"
done
timeout -k 10 200 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_15k.json 2> $O/bench_15k.err || { echo "bench 15k failed"; tail $O/bench_15k.err; exit 1; }
python3 -c "
import json; d=json.load(open('$O/bench_15k.json')); print('15k', d['value'], d['ms_per_step'], d.get('parity'))"
cd /tmp
prof() {   # prof <set> <kind> <bench args...>
  set_=$1; kind=$2; shift 2
  case $kind in
    stats) opts="--kernel-trace --stats"; st="--steps 4 --warmup 1" ;;
    fetch) opts="--pmc FETCH_SIZE"; st="--steps 1 --warmup 1" ;;
    write) opts="--pmc WRITE_SIZE"; st="--steps 1 --warmup 1" ;;
  esac
  timeout -k 10 240 rocprofv3 $opts --output-format csv -d $O/$set_/$kind -- python3 $R/bench.py $st --no-cpu-baseline "$@" > $O/${set_}_$kind.log 2>&1
  rc=$?; echo "$set_ $kind rc=$rc"
  find $O/$set_/$kind -type f ! -name "*kernel_stats.csv" ! -name "*counter_collection.csv" -delete 2>/dev/null
  return $rc
}
for kind in stats fetch write; do prof random100k_k15 $kind --workload random100k -k 15 || exit 1; done
cd $R
bash scripts/gpu/fuzz.sh r5y/fuzz 150
