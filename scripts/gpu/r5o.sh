# round 5, call F: the round's profile set -- kernel stats + HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of the default
# workload, of one GPU's share of random100k at k = 15 / 19 / 31 (configs[4] sweep) and of salmonella500 at w = 10; SQ counters of
# the sketch kernel on the default workload.  Summaries: python3 scripts/summarize_profiles.py gpurun_out/r5o/<set> profiles/r05_<set> ...
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5o; mkdir -p $O; cd /tmp
prof() {   # prof <set> <kind> <steps args...> -- <bench args...>
  set_=$1; kind=$2; shift 2
  case $kind in
    stats) opts="--kernel-trace --stats"; st="--steps 4 --warmup 1" ;;
    pmc)   opts="--pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY"; st="--steps 1 --warmup 1" ;;
    fetch) opts="--pmc FETCH_SIZE"; st="--steps 1 --warmup 1" ;;
    write) opts="--pmc WRITE_SIZE"; st="--steps 1 --warmup 1" ;;
  esac
  timeout -k 10 240 rocprofv3 $opts --output-format csv -d $O/$set_/$kind -- python3 $R/bench.py $st --no-cpu-baseline "$@" > $O/${set_}_$kind.log 2>&1
  rc=$?; echo "$set_ $kind rc=$rc"
  # keep only the summaries' inputs (the merge limit of gpurun_out/ is 64 MiB)
  find $O/$set_/$kind -type f ! -name "*kernel_stats.csv" ! -name "*counter_collection.csv" -delete 2>/dev/null
  return $rc
}
for kind in stats pmc fetch write; do prof bacteria15k $kind || exit 1; done
for k in 19 15 31; do for kind in stats fetch write; do prof random100k_k$k $kind --workload random100k -k $k || exit 1; done; done
for kind in stats fetch write; do prof salmonella500_w10 $kind --workload salmonella500 -w 10 || exit 1; done
du -sh $O
cd $R
for k in 15 19 31; do timeout -k 10 300 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --workload random100k -k $k > $O/bench_random100k_k$k.json 2>$O/bench_random100k_k$k.err; python3 -c "
import json; d=json.load(open('$O/bench_random100k_k$k.json')); print('r100k k$k', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['parity'])"; done
timeout -k 10 300 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --workload salmonella500 -w 10 > $O/bench_salmonella500_w10.json 2>$O/bench_w10.err
timeout -k 10 300 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --workload salmonella500 > $O/bench_salmonella500.json 2>$O/bench_s500.err
python3 -c "
import json
for f in ('bench_salmonella500_w10','bench_salmonella500'):
    d=json.load(open('$O/'+f+'.json')); print(f, d['value'], d['ms_per_step'], d['roofline']['kernel'][:40], d['roofline']['frac'])"
du -sh $O
