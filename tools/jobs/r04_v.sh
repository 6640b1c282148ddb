# Round-4 (v): one utterance at a time: the fused FFN kernel below its batch threshold (1000 rows) -- does it shorten the 99-launch encoder?
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04/v; mkdir -p $O
X="--gpus 1 --steps 4 --warmup 2 --no-cpu-baseline --no-multilingual --no-bf16x3-line --no-streaming-line --no-rccl-probe --no-bracket-ab --no-prof"
run() { tag=$1; shift; env "$@" timeout 600 python bench.py $X > $O/b_$tag.json 2> $O/b_$tag.err; python -c "import json; d=json.load(open('$O/b_$tag.json')); print('$tag:', d['latency_ms_single_stream'], d.get('latency_ms_single_stream_launch_per_op_mt_step'), d.get('rtfx_single_stream'))" || tail -3 $O/b_$tag.err; }
run default A=1
run ffn_min96 SS_FFN_MIN_ROWS=96
run ffn_min48 SS_FFN_MIN_ROWS=48
run ffn_min96_wm4 SS_FFN_MIN_ROWS=96 SS_FFN_WM=4
run default2 A=1
