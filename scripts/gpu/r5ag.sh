# round 5, call AG: sw_graph_export sends nodes and edges packed (12 of 40 / 20 of 24 bytes) and the copiers expand them:
# the download / drop-in tests, then the default bench line twice -- packed and whole (SEQWIN_AMD_EXPORT_WHOLE=1) -- on one box
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5ag; mkdir -p $O; cd $R
timeout -k 10 540 python3 -m pytest tests -q -m gpu -x > $O/tests.log 2>&1; rc=$?; echo "suite rc=$rc"; tail -n 4 $O/tests.log; [ $rc = 0 ] || exit 1
for v in packed whole packed2 whole2; do
  case $v in whole*) export SEQWIN_AMD_EXPORT_WHOLE=1 ;; *) unset SEQWIN_AMD_EXPORT_WHOLE ;; esac
  SEQWIN_BENCH_E2E_NCPU=16,32,128 timeout -k 10 400 python3 bench.py --steps 3 --warmup 1 > $O/bench_$v.json 2> $O/bench_$v.err; echo "bench $v rc=$?"; python3 -c "
import json; d=json.load(open('$O/bench_$v.json')); s=d['e2e']['split_ms']; print('$v', d['e2e']['value'], d['e2e']['by_n_cpu'], 'export', s['export_ms'], 'ingest', s['ingest_upload_ms'], 'penalty', s['get_penalty_wall_ms'], 'total', s['total_wall_ms'], d['e2e']['equal_to_cpu_baseline'])"
done
