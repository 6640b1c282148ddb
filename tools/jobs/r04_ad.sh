# Round-4 (ad): 8-float row padding in the five register-streamed kernels (conflict-free ds_read_b128 lane groups) vs the 4-float padding
# (tools/libss_pad4.so = the build before): op tests, LdsBankConflict counter, micro-benchmarks, bench A/B
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04/ad; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_ffn_gpu.py tests/test_rtlin_gpu.py tests/test_ops_gpu.py tests/test_batch_gpu.py -q -x ) > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for lib in pad4 pad8; do
  [ $lib = pad4 ] && export SS_HIP_LIB=$PWD/tools/libss_pad4.so || unset SS_HIP_LIB
  echo "== $lib: conv_c64 / c32 / c16 slab columns (conv1 | conv2, us)"
  timeout 200 python tools/c64_bench.py 2>&1 | grep -E "^ +(3|7|11) +1 " | awk -F'|' '{print $1 "|" $3 "|" $5}'
  C64_BENCH_CHANNELS=32 timeout 200 python tools/c64_bench.py 2>&1 | grep -E "^ +(3|7|11) +1 " | awk -F'|' '{print $1 "|" $3 "|" $5}'
  C64_BENCH_CHANNELS=16 timeout 200 python tools/c64_bench.py 2>&1 | grep -E "^ +(3|7|11) +1 " | awk -F'|' '{print $1 "|" $3 "|" $5}'
done 2>&1 | tee $O/micro.txt
X="--gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-multilingual --no-bf16x3-line --no-streaming-line --no-rccl-probe --no-bracket-ab --no-latency-pass"
for lib in pad4 pad8 pad4 pad8; do
  [ $lib = pad4 ] && export SS_HIP_LIB=$PWD/tools/libss_pad4.so || unset SS_HIP_LIB
  timeout 600 python bench.py $X > $O/b_$lib.json 2> $O/b_$lib.err; python -c "import json; d=json.load(open('$O/b_$lib.json')); print('$lib:', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline_second_kernel']['frac'])"
done 2>&1 | tee $O/bench_ab.txt
unset SS_HIP_LIB
timeout 300 rocprofv3 --kernel-trace --pmc LdsBankConflict --output-format csv -d $O/pmc_lds -- python bench.py --steps 6 --warmup 1 --streams 1 --no-latency-pass --no-cpu-baseline --no-bf16x3-line --no-multilingual --no-streaming-line --no-bracket-ab --no-rccl-probe > $O/pmc_lds.bench.json 2> $O/pmc_lds.err
python tools/pmc_extra.py $O/pmc_lds.md LdsBankConflict=$(ls $O/pmc_lds/*/*counter_collection.csv | head -1) | head -30
rm -f $O/pmc_lds/*/*counter_collection.csv $O/pmc_lds/*/*kernel_trace.csv
