cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02/host; mkdir -p $O
TIMEFORMAT="wall %R user %U sys %S"
for st in 8 1; do for k in 20 100; do
  echo "streams $st steps $k"
  time ( timeout 600 python bench.py --steps $k --warmup 5 --streams $st --no-latency-pass --no-cpu-baseline > $O/b_${st}_$k.json 2> $O/b_${st}_$k.err )
  python -c "import json; d=json.load(open('$O/b_${st}_$k.json')); print(d['value'], d['ms_per_step'])"
done; done
