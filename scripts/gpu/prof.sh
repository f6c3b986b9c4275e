# full profile set of the default bench (stats + SQ PMC + FETCH + WRITE, separate rocprofv3 runs) + the small-window and
# iid-random stats: bash scripts/gpu/prof.sh <tag>; then locally: python scripts/summarize_profiles.py gpurun_out/<tag>/prof profiles/<tag>_bacteria15k --steps 5
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O; cd $R
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err &&
SEQWIN_BENCH_FORCE_DIST=1 python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline > $O/bench_dist1.json 2> $O/bench_dist1.err &&
cd /tmp &&
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof/stats -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline > $O/prof_stats.log 2>&1 &&
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $O/prof/pmc -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/prof_pmc.log 2>&1 &&
AVAIL=$(rocprofv3 -L 2>/dev/null | tr -c 'A-Za-z0-9_' '\n' | sort -u) &&
PMC2=$(for c in SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VALU SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE; do echo "$AVAIL" | grep -qx $c && echo -n "$c "; done) &&
echo "pmc2 counters: $PMC2" > $O/prof_pmc2.log &&
rocprofv3 --pmc $PMC2 --output-format csv -d $O/prof/pmc2 -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline >> $O/prof_pmc2.log 2>&1 &&
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/prof/fetch -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/prof_fetch.log 2>&1 &&
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/prof/write -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/prof_write.log 2>&1 &&
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_w10/stats -- python3 $R/bench.py --workload salmonella500 -w 10 --steps 4 --warmup 1 --no-cpu-baseline > $O/prof_w10.log 2>&1 &&
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_rand19/stats -- python3 $R/bench.py --workload random100k -k 19 --steps 4 --warmup 1 --no-cpu-baseline > $O/prof_rand19.log 2>&1 &&
SEQWIN_BENCH_FORCE_DIST=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_dist1/stats -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline > $O/prof_dist1.log 2>&1
echo rc=$?
grep "^{" $O/prof_w10.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('w10', d['value'], d['ms_per_step'], d['stages_ms'], d['roofline']['kernel'], d['roofline']['frac'])"
grep "^{" $O/prof_rand19.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('rand19', d['value'], d['ms_per_step'], d['stages_ms'])"
