# Round-4 (af): concurrent streams per GPU at 64 utterances per batch: 8 (default) vs 12 vs 16, alternating on one box
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04/af; mkdir -p $O
X="--gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-multilingual --no-bf16x3-line --no-streaming-line --no-rccl-probe --no-bracket-ab --no-latency-pass --no-prof"
for s in 8 12 16 8 12 16 8 12 16; do
  timeout 600 python bench.py $X --streams $s > $O/b_$s.json 2> $O/b_$s.err; python -c "import json; d=json.load(open('$O/b_$s.json')); print('streams $s:', d['value'], d['ms_per_step'])"
done 2>&1 | tee $O/streams.txt
