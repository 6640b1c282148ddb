# Round-4 (ae): 8-float padding, conv_c64 keeping its 256-row coverage through the 4-float variant where only that fits: tests + bench A/B
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04/ae; mkdir -p $O
( timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_batch_gpu.py -q -x -k "c64 or fused or batch" ) > $O/pytest.log 2>&1; tail -2 $O/pytest.log
X="--gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-multilingual --no-bf16x3-line --no-streaming-line --no-rccl-probe --no-bracket-ab --no-latency-pass"
for lib in pad4 pad8 pad4 pad8 pad4 pad8; do
  [ $lib = pad4 ] && export SS_HIP_LIB=$PWD/tools/libss_pad4.so || unset SS_HIP_LIB
  timeout 600 python bench.py $X > $O/b_$lib.json 2> $O/b_$lib.err; python -c "import json; d=json.load(open('$O/b_$lib.json')); print('$lib:', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline_second_kernel']['frac'])"
done 2>&1 | tee $O/bench_ab.txt
