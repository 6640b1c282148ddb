cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02/quick; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -4
for i in 1 2; do
timeout 300 python bench.py --steps 20 --warmup 5 --no-latency-pass --no-cpu-baseline > $O/b.json 2> $O/b.err
python -c "import json; d=json.load(open('$O/b.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_us'])"
done
