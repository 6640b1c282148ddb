# Round-4 (ag): Winograd F(2,3) form of the 64-channel stage convs (conv_c64w.hip): op tests, micro-benchmark, bench A/B
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04/ag; mkdir -p $O
( timeout 600 python -m pytest tests/test_ops_gpu.py -q -x -k "conv_c64_slab_kernel" ) > $O/pytest_c64.log 2>&1; tail -12 $O/pytest_c64.log
for w in 0 1; do
  echo "== SS_CONV_C64_WINOGRAD=$w"
  SS_CONV_C64_WINOGRAD=$w timeout 200 python tools/c64_bench.py 2>&1 | grep -E "^ +(3|7|11) +(1|5) " | awk -F'|' '{print $1 "|" $3 "|" $5}'
done | tee $O/micro.txt
