# round 5, call X: host ingest with the packer that crosses line ends and the block-wise read, against the r05a forms, on the GPU box's
# host (16 CPUs of quota): sw_host_ingest alone (tests/tools/host_ingest_time.py), then the default bench line (its e2e leg)
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5x; mkdir -p $O; cd $R
# (first: where k_unsort_adj's 21 ms at random100k k = 15 go -- no candidates at all / no digit counts in LDS, against the A/B build as it is)
cd /tmp
kx() { name=$1; lib=$2; shift 2
  env SEQWIN_AMD_LIB=$R/ab_live/$lib "$@" timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/k15_$name -- python3 $R/bench.py --workload random100k -k 15 --steps 2 --warmup 1 --no-cpu-baseline > $O/k15_$name.log 2>&1
  echo "k15 $name rc=$?"; find $O/k15_$name -type f ! -name "*kernel_stats.csv" -delete 2>/dev/null
  grep -h -E "k_unsort_adj|k_edges_runs|k_rs_count|rocprim.*histogram" $O/k15_$name/*/*kernel_stats.csv | cut -c1-200
}
kx asis libseqwin_hip_ab.so A=1
kx nohist libseqwin_hip_ab.so SEQWIN_AMD_NO_ADJ_HIST=1
kx nocand libseqwin_hip_nocand.so A=1
cd $R
run() { name=$1; shift; env "$@" timeout -k 10 200 python3 tests/tools/host_ingest_time.py 1024 16 32 128 > $O/ingest_$name.txt 2>&1; echo "== $name rc=$?"; cat $O/ingest_$name.txt; }
run default A=1
run block0 SEQWIN_AMD_READ_BLOCK_KB=0
run r05a SEQWIN_AMD_READ_BLOCK_KB=0 SEQWIN_AMD_LINE_PACKER=1
run block64 SEQWIN_AMD_READ_BLOCK_KB=64
run block1024 SEQWIN_AMD_READ_BLOCK_KB=1024
timeout -k 10 400 python3 bench.py --steps 10 --warmup 2 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; python3 -c "
import json; d=json.load(open('$O/bench_default.json')); print(d['value'], d['ms_per_step'], d['e2e']['value'], d['e2e']['by_n_cpu'], d['e2e']['split_ms'], d['e2e']['gz']['routes'], d['parity']['equal'])"
