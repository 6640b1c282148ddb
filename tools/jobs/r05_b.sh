# Round-5 (b): what pack-invariant arithmetic costs -- alternating A/B of SS_PACK_INVARIANT on one box (headline only),
# plus the fused FFN's tile-height cost table for the whole-tile form
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05/b; mkdir -p $O
X="--gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-multilingual --no-bf16x3-line --no-streaming-line --no-rccl-probe --no-bracket-ab --no-latency-pass"
run() { tag=$1; shift; env "$@" timeout 600 python bench.py $X > $O/b_$tag.json 2> $O/b_$tag.err; python -c "import json; d=json.load(open('$O/b_$tag.json')); print('$tag:', d['value'], d['utterances_per_sec'], d['ms_per_step'], d['roofline']['frac'])" || tail -3 $O/b_$tag.err; }
run inv1 SS_PACK_INVARIANT=1
run free1 SS_PACK_INVARIANT=0
run inv2 SS_PACK_INVARIANT=1
run free2 SS_PACK_INVARIANT=0
run inv3 SS_PACK_INVARIANT=1
run free3 SS_PACK_INVARIANT=0
