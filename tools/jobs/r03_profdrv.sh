cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03/profdrv; mkdir -p $O
export TMPDIR=/tmp
for v in "a:--no-multilingual --no-streaming-line" "b:--no-streaming-line" "c:--no-multilingual"; do
  tag=${v%%:*}; flags=${v#*:}
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$tag -- python bench.py --steps 20 --warmup 5 $flags > $O/bench_$tag.json 2> $O/prof_$tag.err
  echo "== $tag ($flags): rc=$? json bytes $(stat -c %s $O/bench_$tag.json)"; grep -v "^    @" $O/prof_$tag.err | grep -i "sigsegv\|error\|abort" | head -3
  rm -f $O/prof_$tag/*/*kernel_trace.csv
done
