R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2n; mkdir -p $O; cd $R
for k in 15 19 31; do
  timeout -k 10 200 python3 bench.py --workload random100k -k $k --steps 3 --warmup 1 --no-cpu-baseline > $O/random100k_k$k.json 2> $O/random100k_k$k.err
  python3 -c "
import json; d=json.load(open('$O/random100k_k$k.json')); print('k=$k', d['value'], d['ms_per_step'], d['counts'], {a:round(b,1) for a,b in d['stages_ms'].items()}, d['plan_ms'])" || tail -3 $O/random100k_k$k.err
done
for w in 50 10; do
  timeout -k 10 200 python3 bench.py --workload salmonella500 -w $w --steps 5 --warmup 1 --no-cpu-baseline > $O/salm_w$w.json 2> $O/salm_w$w.err
  python3 -c "
import json; d=json.load(open('$O/salm_w$w.json')); print('w=$w', d['value'], d['ms_per_step'], d['counts'], {a:round(b,1) for a,b in d['stages_ms'].items()})" || tail -3 $O/salm_w$w.err
done
