# Round-5: cost of the pack-invariant routes at the measured configuration (packs of 128), alternating on one box
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05/ab_pi; mkdir -p $O
X="--steps 20 --warmup 5 --no-cpu-baseline --no-multilingual --no-bf16x3-line --no-streaming-line --no-rccl-probe --no-bracket-ab --no-latency-pass --no-soak"
for i in 1 2; do
  SS_PACK_INVARIANT=0 timeout 300 python bench.py $X > $O/off_$i.json 2> $O/off_$i.err; python -c "import json; d=json.load(open('$O/off_$i.json')); print('pack-invariant OFF:', d['value'], d['ms_per_step'], d['pack_invariance']['pack_invariant_context'])"
  timeout 300 python bench.py $X > $O/on_$i.json 2> $O/on_$i.err; python -c "import json; d=json.load(open('$O/on_$i.json')); print('pack-invariant ON :', d['value'], d['ms_per_step'], d['pack_invariance']['alone_equals_in_pack_bitwise'])"
done
