#!/bin/bash
# .fa.gz route of the host ingest, two files per worker against one (tests/tools/e2e_gz_pairs_ab.py), the gz tests of the GPU suite and a
# differential fuzz leg with every input gzipped (host route: the pair decoder), all on the TEST library.
#   bash scripts/gpu/gzpairs.sh <tag> [genomes] [fuzz seconds]
set -eo pipefail
TAG=${1:-gzp}; G=${2:-2400}; FZ=${3:-150}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export SEQWIN_AMD_LIB=$PWD/seqwin_amd/libseqwin_hip_test.so
timeout -k 10 400 python3 tests/tools/e2e_gz_pairs_ab.py $G 64 3 > $OUT/gz_pairs_ab.txt 2>&1 && cat $OUT/gz_pairs_ab.txt &&
timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "device_gz_ingest or streaming_ingest or fuzz" > $OUT/pytest_gz.txt 2>&1 && tail -3 $OUT/pytest_gz.txt &&
{ export SEQWIN_AMD_GZ_PAIRS=1; FUZZ_GZ=1 timeout -k 10 $((FZ+60)) python3 tests/tools/fuzz_gpu.py $FZ 61 > $OUT/fuzz_host_gz_a.log 2>&1 &
  FUZZ_GZ=1 timeout -k 10 $((FZ+60)) python3 tests/tools/fuzz_gpu.py $FZ 62 > $OUT/fuzz_host_gz_b.log 2>&1 &
  FUZZ_GZ=1 SEQWIN_AMD_INGEST_WINDOW=1 timeout -k 10 $((FZ+60)) python3 tests/tools/fuzz_gpu.py $FZ 63 > $OUT/fuzz_host_gz_window1.log 2>&1 &
  for i in $(seq 1 20); do sleep 30; echo "t=$((i*30))s"; [ $((i*30)) -ge $FZ ] && break; done; wait; tail -n 2 $OUT/fuzz_host_gz_*.log; }
