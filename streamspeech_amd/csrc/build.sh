#!/bin/bash
# Build libstreamspeech_hip.so for gfx950 (in-tree; the .so travels to the GPU box with the snapshot).
# Tuning builds: SS_EXTRA_FLAGS="-DK2_SCHED=0" SS_BUILD_DIR=build/k2s0 SS_OUT_LIB=../../tools/libss_k2s0.so bash build.sh
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-result ${SS_EXTRA_FLAGS:-}"
OUT=${SS_OUT_LIB:-../libstreamspeech_hip.so}
B=${SS_BUILD_DIR:-build}
SRCS="gemm conv_sk conv_sk2 conv_slab resblock ffn rtlin conv_c64 conv_c64w conv_c32 conv_c16 attention elementwise fbank mt_step enc_step model batch vocoder debug_ops"
mkdir -p $B
rm -f $B/.failed          # stale flag of an earlier run: cleared BEFORE the jobs start (a fast failure must survive)
for f in $SRCS; do
  if [ ! -f $B/$f.o ] || [ $f.hip -nt $B/$f.o ] || [ -n "$(find . -maxdepth 1 -name '*.hpp' -newer $B/$f.o)" ] \
     || [ ../../include/streamspeech_hip.h -nt $B/$f.o ]; then
    ( $HIPCC $FLAGS -c $f.hip -o $B/$f.o.tmp && mv $B/$f.o.tmp $B/$f.o ) || { rm -f $B/$f.o $B/$f.o.tmp; touch $B/.failed; } &
  fi
done
wait
if [ -f $B/.failed ]; then echo "build FAILED"; rm -f $B/.failed; exit 1; fi
OBJS=""
for f in $SRCS; do OBJS="$OBJS $B/$f.o"; done
$HIPCC --offload-arch=gfx950 -shared -fPIC $OBJS -o $OUT
python3 -c "import ctypes,sys; ctypes.CDLL(sys.argv[1])" "$(realpath $OUT)" && echo "built $(realpath $OUT)"
