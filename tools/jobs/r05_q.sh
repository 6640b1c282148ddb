# Round-5 (q): pack size sweep on the current build (ids cannot depend on it any more: pack-invariant arithmetic)
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05/q; mkdir -p $O
X="--gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-multilingual --no-bf16x3-line --no-streaming-line --no-rccl-probe --no-bracket-ab --no-latency-pass --no-soak"
run() { tag=$1; shift; timeout 600 python bench.py $X "$@" > $O/b_$tag.json 2> $O/b_$tag.err; python -c "import json; d=json.load(open('$O/b_$tag.json')); print('$tag:', d['value'], d['utterances_per_sec'], d['ms_per_step'])" || tail -3 $O/b_$tag.err; }
run b64a --batch 64
run b96a --batch 96
run b128a --batch 128
run b64b --batch 64
run b96b --batch 96
run b128b --batch 128
run b64s12 --batch 64 --streams 12
run b48 --batch 48
