#!/bin/bash
# Build libstreamspeech_hip.so for gfx950 (in-tree; the .so travels to the GPU box with the snapshot).
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-result"
mkdir -p build
rm -f build/.failed          # stale flag of an earlier run: cleared BEFORE the jobs start (a fast failure must survive)
for f in gemm conv_sk conv_slab attention elementwise fbank model; do
  if [ ! -f build/$f.o ] || [ $f.hip -nt build/$f.o ] || [ -n "$(find . -maxdepth 1 -name '*.hpp' -newer build/$f.o)" ] \
     || [ ../../include/streamspeech_hip.h -nt build/$f.o ]; then
    ( $HIPCC $FLAGS -c $f.hip -o build/$f.o.tmp && mv build/$f.o.tmp build/$f.o ) || { rm -f build/$f.o build/$f.o.tmp; touch build/.failed; } &
  fi
done
wait
if [ -f build/.failed ]; then echo "build FAILED"; rm -f build/.failed; exit 1; fi
$HIPCC --offload-arch=gfx950 -shared -fPIC build/gemm.o build/conv_sk.o build/conv_slab.o build/attention.o build/elementwise.o build/fbank.o build/model.o \
  -o ../libstreamspeech_hip.so
python3 -c "import ctypes,sys; ctypes.CDLL(sys.argv[1])" "$(realpath ../libstreamspeech_hip.so)" && echo "built $(realpath ../libstreamspeech_hip.so)"
