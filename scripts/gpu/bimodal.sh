# bimodal.sh <tag> [runs]: the default build in N fresh processes with the addresses of every large pool block logged -- which
# placement goes with the 47 ms and which with the 50.5 ms nodes stage (NOTES.md, r05: "the nodes stage is bimodal between processes")
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O; cd $R; N=${2:-8}
for i in $(seq 1 $N); do
  SEQWIN_AMD_DEBUG_ALLOC=1 $BIMODAL_ENV timeout -k 10 120 python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/run$i.json 2> $O/run$i.err || exit 1
  python3 - $O/run$i.json $O/run$i.err <<'PY'
import json, re, sys
d = json.load(open(sys.argv[1]))
addrs = re.findall(r"hipMalloc ([0-9.]+) GiB at (0x[0-9a-f]+)", open(sys.argv[2]).read())
s = d["stages_ms"]
print(f"{d['ms_per_step']:.2f} ms  sketch {s['sketch_ms']:.2f} nodes {s['nodes_ms']:.2f} edges {s['edges_ms']:.2f}  blocks:", " ".join(f"{g}@{a[-9:]}" for g, a in addrs[:14]))
PY
done
