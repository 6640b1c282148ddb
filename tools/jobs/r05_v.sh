# Round-5 (v): epilogue requests both rows' residual operand per pair tile, no weight-ring prefetch across the block boundary (CW_RPREF) vs the committed build
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05/v; mkdir -p $O
for ch in 128 64 256; do
  C64_BENCH_CHANNELS=$ch timeout 300 python tools/c64_bench.py > $O/micro_c${ch}_new.txt 2>&1
  SS_HIP_LIB=tools/bin/libss_prev.so C64_BENCH_CHANNELS=$ch timeout 300 python tools/c64_bench.py > $O/micro_c${ch}_prev.txt 2>&1
  echo "== $ch channels: conv1 | conv2+R (new), conv1 | conv2+R (committed)"
  paste <(awk -F'|' 'NR>3{print $1 "|" $3 "|" $5}' $O/micro_c${ch}_new.txt) <(awk -F'|' 'NR>3{print $3 "|" $5}' $O/micro_c${ch}_prev.txt) | grep -v "^(the" | sed 's/([^)]*)//g'
done
