#!/bin/bash
# build_variant.sh <name> <extra hipcc flags...>: an A/B build of libseqwin_hip.so under ab_live/ (git-ignored, shipped to the GPU box: delete it after the call that used it -- every gpurun pushes it);
# select it with SEQWIN_AMD_LIB=ab_live/libseqwin_hip_<name>.so
set -e
R=$(cd "$(dirname "$0")/../.." && pwd); name=$1; shift
B=$R/seqwin_amd/csrc/build_$name; mkdir -p $B $R/ab_live
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-result -I$R/include -ffp-contract=off $*"
for f in sketch index radix ingest_dev multi api; do /opt/rocm/bin/hipcc $F -c $R/seqwin_amd/csrc/$f.hip -o $B/$f.o & done
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -I$R/include -c $R/seqwin_amd/csrc/host_ingest.cpp -o $B/host_ingest.o
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $B/*.o -o $R/ab_live/libseqwin_hip_$name.so -lz -lpthread -Wl,-rpath,/opt/rocm/lib
rm -rf $B
echo built ab_live/libseqwin_hip_$name.so
