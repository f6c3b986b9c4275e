# round 4, call U: pipelined download (ring of pinned slots + copying threads) against one hipMemcpy per array
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4u; mkdir -p $O; cd $R
cat /sys/kernel/mm/transparent_hugepage/enabled /sys/kernel/mm/transparent_hugepage/defrag; nproc
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "pipelined_download" > $O/tests.log 2>&1; rc=$?; tail -n 3 $O/tests.log
[ $rc -eq 0 ] || exit $rc
export SEQWIN_AMD_DOWNLOAD_STREAMS=1
for v in 0 32 128 512 0 128; do
  if [ $v = 0 ]; then unset SEQWIN_AMD_DOWNLOAD_KERNEL; else export SEQWIN_AMD_DOWNLOAD_KERNEL=$v; fi
  timeout -k 10 400 python3 bench.py --steps 3 --warmup 1 > $O/bench_$v.json 2> $O/bench_$v.err || exit 1
  python3 -c "import json; d=json.loads([l for l in open('$O/bench_$v.json') if l.startswith('{')][-1]); e=d['e2e']; print('$v', e['value'], e['equal_to_cpu_baseline'], e['split_ms']['export_ms'], e['split_ms']['total_wall_ms'])"
done
