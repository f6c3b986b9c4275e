# round 5, call AK: kmers cross PCIe in 5 bytes (pos | record_idx << pos_bits) as nodes and edges in 12 / 20: the GPU suite, then the
# default bench line's e2e leg packed / kmers whole / all whole, alternating on one box
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5ak; mkdir -p $O; cd $R
timeout -k 10 560 python3 -m pytest tests -q -m gpu -x > $O/tests.log 2>&1; rc=$?; echo "suite rc=$rc"; tail -n 4 $O/tests.log; [ $rc = 0 ] || exit 1
for v in packed kwhole whole packed2 kwhole2; do
  unset SEQWIN_AMD_EXPORT_WHOLE SEQWIN_AMD_EXPORT_KMERS_WHOLE
  case $v in whole*) export SEQWIN_AMD_EXPORT_WHOLE=1 ;; kwhole*) export SEQWIN_AMD_EXPORT_KMERS_WHOLE=1 ;; esac
  SEQWIN_BENCH_E2E_NCPU=16,32,128 timeout -k 10 400 python3 bench.py --steps 3 --warmup 1 > $O/bench_$v.json 2> $O/bench_$v.err; echo "bench $v rc=$?"; python3 -c "
import json; d=json.load(open('$O/bench_$v.json')); s=d['e2e']['split_ms']; print('$v', d['e2e']['value'], d['e2e']['by_n_cpu'], 'export', s['export_ms'], 'ingest', s['ingest_upload_ms'], 'penalty', s['get_penalty_wall_ms'], 'total', s['total_wall_ms'], 'cpu', s['cpu_s_build_export_penalty'], d['e2e']['equal_to_cpu_baseline'])"
done
