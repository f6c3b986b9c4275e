R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2k; mkdir -p $O; cd $R
python3 bench.py --genomes 1500 --steps 2 --warmup 1 --no-cpu-baseline > $O/n1.json 2> $O/n1.err
SEQWIN_BENCH_BACKEND=gloo timeout -k 10 400 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --genomes 1500 --steps 2 --warmup 1 > $O/n2.json 2> $O/n2.err
SEQWIN_BENCH_BACKEND=gloo timeout -k 10 400 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 3 --master-addr 127.0.0.1 --master-port 29612 bench.py --gpus 3 --genomes 1500 --steps 2 --warmup 1 > $O/n3.json 2> $O/n3.err
python3 - <<PY
import json
for f in ("n1","n2","n3"):
    try:
        d=[json.loads(l) for l in open("$O/%s.json"%f) if l.startswith("{")][0]
        print(f, d["n_gpus"], d["value"], d["ms_per_step"], d["counts"], d["checksums"], {k:round(v,1) for k,v in d["stages_ms"].items()})
    except Exception as e:
        print(f, "FAILED", e); print(open("$O/%s.err"%f).read()[-1500:])
PY
timeout -k 10 300 python3 tests/tools/e2e_fasta.py 256 > $O/e2e.log 2>&1; tail -12 $O/e2e.log
