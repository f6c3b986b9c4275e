# round 4, call X: e2e ingest with the 64-byte (AVX-512) packer against the 32-byte (AVX2) one
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4x; mkdir -p $O; cd $R
for v in avx2 avx512 avx2 avx512; do
  if [ $v = avx2 ]; then export SEQWIN_AMD_NO_AVX512=1; else unset SEQWIN_AMD_NO_AVX512; fi
  SEQWIN_AMD_DEBUG_TIMING=1 timeout -k 10 400 python3 bench.py --steps 2 --warmup 1 > $O/bench_$v.json 2> $O/bench_$v.err || exit 1
  python3 -c "import json; d=json.loads([l for l in open('$O/bench_$v.json') if l.startswith('{')][-1]); e=d['e2e']; print('$v', e['value'], e['equal_to_cpu_baseline'], e['split_ms']['ingest_upload_ms'], e['split_ms']['get_penalty_wall_ms'], e['split_ms']['total_wall_ms'], e['sample'][-50:])"
  grep "sink thread waited\|workers: parse" $O/bench_$v.err | tail -n 4
done
