# Round-4 (aj): 128-channel stage on the Winograd slab kernel (no twins in that stage) vs conv_sk2<128>: vocoder tests + bench A/B
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04/aj; mkdir -p $O
( timeout 900 python -m pytest tests/test_batch_gpu.py tests/test_stages_gpu.py tests/test_multilingual_gpu.py -q -x -k "vocoder or fused or batch_mt_t2u or multilingual or offline" ) > $O/pytest.log 2>&1; tail -3 $O/pytest.log
X="--gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-multilingual --no-bf16x3-line --no-streaming-line --no-rccl-probe --no-bracket-ab --no-latency-pass"
for w in 0 1 0 1 0 1; do
  SS_CONV_C128_WINOGRAD=$w timeout 600 python bench.py $X > $O/b_$w.json 2> $O/b_$w.err; python -c "import json; d=json.load(open('$O/b_$w.json')); print('c128 winograd $w:', d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline_second_kernel']['kernel'], d['roofline_second_kernel']['frac'])" || tail -3 $O/b_$w.err
done 2>&1 | tee $O/bench_ab.txt
