# Round-4 (z3): ping-pong schedule with a 16-fragment weight ring (a whole tap ahead): full / pure contraction (dbg 6) vs the two-workgroup form
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04/z; mkdir -p $O
for cfg in "0 0" "1 0" "1 6" "1 2" "1 4"; do
  set -- $cfg
  echo "== SS_CONV_C64_PP=$1 SS_CONV_C64_PP_DBG=$2"
  SS_CONV_C64_PP=$1 SS_CONV_C64_PP_DBG=$2 timeout 200 python tools/c64_bench.py 2>&1 | grep -E "^ +(3|7|11) +1 "
done | tee $O/pp_ring16.txt
