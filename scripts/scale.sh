#!/bin/bash
# scale.sh -- the 1 / 2 / 4 / 8-GPU lines of the default bench on ONE node, both hosts of the sharded build:
#   (a) one process per GPU, torch.distributed over RCCL:  bench.py --gpus N   (what the driver's SCALE runs launch)
#   (b) one process, one host thread per GPU inside the library (SEQWIN_DEVICES, csrc/multi.hip): FASTA in /dev/shm -> sw_build
# Every line carries parity.n1_checksums_equal (the N slices' checksum shares add up to the reference-derived single-GPU values)
# and dist.{distinct_gpus, peer_access, collectives_checked}; bench.py refuses to time a run that is not a scaling point.
# usage: bash scripts/scale.sh [outdir] [max_gpus]      (needs a node with that many GPUs; never run from a one-GPU gpurun box)
set -u
cd "$(dirname "$0")/.."
OUT=${1:-gpurun_out/scale}; MAXN=${2:-8}; mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
NDEV=$(python3 -c "import torch; print(torch.cuda.device_count())")
echo "devices visible: $NDEV" | tee "$OUT/scale.log"
for N in 1 2 4 8; do
  [ "$N" -le "$MAXN" ] && [ "$N" -le "$NDEV" ] || continue
  PORT=$((29600 + N))
  if [ "$N" -eq 1 ]; then
    python3 bench.py --gpus 1 --steps 10 --warmup 2 --no-cpu-baseline > "$OUT/rccl_n$N.json" 2> "$OUT/rccl_n$N.err"
  else
    python3 -m torch.distributed.run --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 --master-port "$PORT" \
      bench.py --gpus "$N" --steps 10 --warmup 2 > "$OUT/rccl_n$N.json" 2> "$OUT/rccl_n$N.err"
  fi
  echo "rccl N=$N rc=$?" | tee -a "$OUT/scale.log"
  python3 - "$OUT/rccl_n$N.json" <<'PY' | tee -a "$OUT/scale.log"
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print("  ", d.get("value"), "Gbp/s", d.get("ms_per_step"), "ms/step; parity", d.get("parity"), "; distinct", (d.get("dist") or {}).get("distinct_gpus"),
          "peer", (d.get("dist") or {}).get("peer_access"), "refused:", d.get("refused"))
except Exception as e:
    print("   no line:", e)
PY
done
# (b) inside one sw_build: the same genomes as FASTA, SEQWIN_DEVICES = the first N devices
python3 tests/tools/multi_device_scale.py "$OUT" "$MAXN" 2>&1 | tee -a "$OUT/scale.log"
