cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05/g; mkdir -p $O
C64_BENCH_CHANNELS=256 timeout 300 python tools/c64_bench.py > $O/c256_micro_npc10.txt 2>&1; cat $O/c256_micro_npc10.txt
C64_BENCH_CHANNELS=128 timeout 300 python tools/c64_bench.py > $O/c128_micro.txt 2>&1; cat $O/c128_micro.txt
