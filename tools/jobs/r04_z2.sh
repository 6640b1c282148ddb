# Round-4 (z2): decomposition of the ping-pong schedule: no setprio (1), no epilogue (2), no re-staging (4) -- timing only
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04/z; mkdir -p $O
for d in 0 1 2 4 6 7; do
  echo "== SS_CONV_C64_PP_DBG=$d"
  SS_CONV_C64_PP_DBG=$d timeout 200 python tools/c64_bench.py 2>&1 | grep -E "^ +(3|7|11) +1 "
done | tee $O/pp_decomposition.txt
