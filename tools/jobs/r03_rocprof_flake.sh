# how often does rocprofv3 survive the driver's bench command?  (4 tries, side legs off to keep it short)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r03/flake; mkdir -p $O
X="--no-cpu-baseline --no-bf16x3-line --no-multilingual --no-streaming-line --no-latency-pass"
for i in 1 2 3 4; do
  rm -rf $O/p$i
  if rocprofv3 --kernel-trace --stats --output-format csv -d $O/p$i -- python bench.py --steps 20 --warmup 5 $X > $O/b$i.json 2> $O/e$i.err; then echo "try $i ok $(python -c "import json; print(json.load(open('$O/b$i.json'))['value'])")"; else echo "try $i CRASHED"; fi
  rm -f $O/p$i/*/*kernel_trace.csv
done
