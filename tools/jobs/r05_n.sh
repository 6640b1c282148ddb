# Round-5 (n): tile-height table of the pack-invariant fused FFN; GPU suite on the dispatch-settings refactor + C32 k >= 7
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05/n; mkdir -p $O
timeout 600 python tools/ffn_canon_bench.py > $O/ffn_canon_bench.txt 2>&1; cat $O/ffn_canon_bench.txt
timeout 2400 python -m pytest tests -q -m gpu -x --deselect tests/test_bench_config_gpu.py > $O/pytest_gpu.log 2>&1; tail -6 $O/pytest_gpu.log
