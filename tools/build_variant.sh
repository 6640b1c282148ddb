#!/bin/bash
# Tuning builds of the library with -DSK_VARIANT=<n> (conv_sk.hip main-loop variants) -> tools/libss_var<n>.so;
# run a tool against one with SS_HIP_LIB=tools/libss_var<n>.so python tools/conv_bench.py sk
set -e
N=${1:?variant number}
cd "$(dirname "$0")/../streamspeech_amd/csrc"
mkdir -p build/var$N
for f in gemm conv_sk conv_slab attention elementwise fbank model; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DSK_VARIANT=$N -c $f.hip -o build/var$N/$f.o 2>/dev/null &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build/var$N/*.o -o ../../tools/libss_var$N.so
echo "built tools/libss_var$N.so"
