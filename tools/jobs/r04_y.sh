# Round-4 (y): which workgroups of conv_c64 share a CU, and does a phase offset between them pay?  (delay = 10 us on bit b of the id)
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04/y; mkdir -p $O
for m in 0 1 2 3 5 10; do
  echo "== SS_CONV_C64_STAGGER=$m"
  SS_CONV_C64_STAGGER=$m timeout 200 python tools/c64_bench.py 2>&1 | grep -E "^ +(3|7|11) +1 "
done | tee $O/stagger_modes.txt
