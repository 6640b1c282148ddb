# Round-5 (k): the incomplete last tap group as F(2,1) / F(2,2) (2 / 3 products instead of a zero-padded F(2,3)'s 4): op tests, micro at
# every width, bench A/B against the previous build (tools/bin/libss_prev.so)
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05/k; mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py -q -m gpu -k "c256 or c128 or c64_slab" > $O/ops.log 2>&1; tail -5 $O/ops.log
for ch in 256 128 64 32; do
  C64_BENCH_CHANNELS=$ch timeout 300 python tools/c64_bench.py > $O/micro_c${ch}_new.txt 2>&1
  SS_HIP_LIB=tools/bin/libss_prev.so C64_BENCH_CHANNELS=$ch timeout 300 python tools/c64_bench.py > $O/micro_c${ch}_prev.txt 2>&1
  echo "== $ch channels: conv1 | conv2 (new), conv1 | conv2 (prev)"
  paste <(awk -F'|' 'NR>3{print $1 "|" $3 "|" $5}' $O/micro_c${ch}_new.txt) <(awk -F'|' 'NR>3{print $3 "|" $5}' $O/micro_c${ch}_prev.txt) | grep -v "^(the"
done
X="--gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-multilingual --no-bf16x3-line --no-streaming-line --no-rccl-probe --no-bracket-ab --no-latency-pass --no-soak"
run() { tag=$1; shift; env "$@" timeout 600 python bench.py $X > $O/b_$tag.json 2> $O/b_$tag.err; python -c "import json; d=json.load(open('$O/b_$tag.json')); print('$tag:', d['value'], d['ms_per_step'])" || tail -3 $O/b_$tag.err; }
run n1 A=1
run p1 SS_HIP_LIB=tools/bin/libss_prev.so
run n2 A=1
run p2 SS_HIP_LIB=tools/bin/libss_prev.so
run n3 A=1
run p3 SS_HIP_LIB=tools/bin/libss_prev.so
