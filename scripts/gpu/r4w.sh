# round 4, call W: k_unsort_adj variants under rocprofv3 (kernel averages of a 4-step bench)
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4w; mkdir -p $O
for v in fused; do
  if [ $v = nohist ]; then export SEQWIN_AMD_UADJ_NOHIST=1; else unset SEQWIN_AMD_UADJ_NOHIST; fi
  cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$v -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline > $O/prof_$v.log 2>&1
  f=$(ls $O/prof_$v/*/*kernel_stats.csv | tail -n 1); python3 - "$f" $v <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r['TotalDurationNs']) for r in rows if 'k_synth' not in r['Name'] and 'k_checksum' not in r['Name'])/5e6
sel=[(r['Name'].split('(anonymous namespace)::')[-1][:24], round(float(r['AverageNs'])/1e3,1)) for r in rows if any(k in r['Name'] for k in ('k_unsort','k_adj','k_pen_bits','stage_prepare','k_rle','k_edges_runs','pair_pass','k_nodes'))]
print(sys.argv[2], round(tot,2), sel)
PY
done
