cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02; mkdir -p $O
SS_SHAPE_LOG=$O/shapes.txt timeout 300 python bench.py --steps 6 --warmup 1 --streams 1 --no-latency-pass --no-cpu-baseline > $O/shapes_bench.json 2> $O/shapes.err
sort -k1n -k9nr $O/shapes.txt | head -100
