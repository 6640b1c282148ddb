# Round-4 (ai): the Winograd slab kernel at 128 channels (8-wave workgroup, two column sets): op tests, micro-benchmark vs conv_sk2<128>
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04/ai; mkdir -p $O
( timeout 600 python -m pytest tests/test_ops_gpu.py -q -x -k "conv_c128_winograd or conv_c64_slab_kernel" ) > $O/pytest.log 2>&1; tail -4 $O/pytest.log
C64_BENCH_CHANNELS=128 timeout 300 python tools/c64_bench.py 2>&1 | grep -E "^ +(3|7|11) +(1|3|5) |rows" | tee $O/micro128.txt
