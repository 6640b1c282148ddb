# Round-4 (ah): Winograd form of the 64-channel stage: micro-benchmark incl. dilation 3, tests that run the vocoder, bench A/B (env knob)
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04/ah; mkdir -p $O
( timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_batch_gpu.py tests/test_stages_gpu.py -q -x -k "conv_c64_slab_kernel or vocoder or fused or batch_mt_t2u" ) > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for w in 0 1; do
  echo "== SS_CONV_C64_WINOGRAD=$w"
  SS_CONV_C64_WINOGRAD=$w timeout 200 python tools/c64_bench.py 2>&1 | grep -E "^ +(3|7|11) +(1|3|5) " | awk -F'|' '{print $1 "|" $3 "|" $5}'
done | tee $O/micro.txt
X="--gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-multilingual --no-bf16x3-line --no-streaming-line --no-rccl-probe --no-bracket-ab --no-latency-pass"
for w in 0 1 0 1 0 1; do
  SS_CONV_C64_WINOGRAD=$w timeout 600 python bench.py $X > $O/b_$w.json 2> $O/b_$w.err; python -c "import json; d=json.load(open('$O/b_$w.json')); print('winograd $w:', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline_second_kernel']['kernel'], d['roofline_second_kernel']['frac'])"
done 2>&1 | tee $O/bench_ab.txt
