#!/bin/bash
# Timing-only ablation build of the kernels (-DSS_ABLATE): tools/libss_ablate.so, used by tools/ablate_bench.py.
set -e
cd "$(dirname "$0")/../streamspeech_amd/csrc"
mkdir -p build/ablate
for f in gemm conv_sk conv_slab attention elementwise fbank model; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DSS_ABLATE -c $f.hip -o build/ablate/$f.o 2>/dev/null &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build/ablate/*.o -o ../../tools/libss_ablate.so
echo "built tools/libss_ablate.so"
