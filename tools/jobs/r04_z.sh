# Round-4 (z): conv_c64 ping-pong schedule (one 8-wave workgroup per CU, halves one phase apart) vs two independent workgroups per CU
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04/z; mkdir -p $O
( timeout 600 python -m pytest tests/test_ops_gpu.py -q -x -k "conv_c64_slab_kernel and 64-" ) > $O/pytest_c64.log 2>&1; tail -4 $O/pytest_c64.log
for pp in 0 1; do
  echo "== SS_CONV_C64_PP=$pp"
  SS_CONV_C64_PP=$pp timeout 200 python tools/c64_bench.py 2>&1 | grep -E "^ +(3|7|11) +(1|5) "
done | tee $O/pp_vs_two_wg.txt
