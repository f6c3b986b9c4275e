# round 4, call Q: fast sketch class for 4 <= w < 16 (runs of 8 / 4): parity, then salmonella500 at w = 10 / 5 before and after
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4q; mkdir -p $O; cd $R
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q > $O/tests_parity.log 2>&1; rc=$?; echo "parity rc=$rc"; tail -n 12 $O/tests_parity.log
[ $rc -eq 0 ] || exit 1
for w in 10 5 15; do for mode in nosmall default; do
  if [ $mode = nosmall ]; then export SEQWIN_AMD_SKETCH=nosmall; else unset SEQWIN_AMD_SKETCH; fi
  timeout -k 10 300 python3 bench.py --workload salmonella500 -w $w --steps 4 --warmup 1 --no-cpu-baseline > $O/b_${w}_$mode.json 2> $O/b_${w}_$mode.err; python3 -c "
import json; d=json.load(open('$O/b_${w}_$mode.json')); print('w=$w $mode', d['value'], d['ms_per_step'], d['stages_ms'], d['checksums'][0][:8], d['counts'])"; done; done
unset SEQWIN_AMD_SKETCH
timeout -k 10 300 python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; python3 -c "
import json; d=json.load(open('$O/bench.json')); print(d['value'], d['ms_per_step'], d['stages_ms'], d.get('parity'))"
