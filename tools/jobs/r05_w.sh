# Round-5 (w): s_setprio(2) during the contraction of the Winograd slab kernels (libss_prio.so) vs the committed build: micro + bench pairs
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05/w; mkdir -p $O
for ch in 64 128; do
  SS_HIP_LIB=tools/bin/libss_prio.so C64_BENCH_CHANNELS=$ch timeout 300 python tools/c64_bench.py > $O/micro_c${ch}_prio.txt 2>&1
  C64_BENCH_CHANNELS=$ch timeout 300 python tools/c64_bench.py > $O/micro_c${ch}_base.txt 2>&1
  echo "== $ch channels: conv1 | conv2+R (prio 2), conv1 | conv2+R (committed)"
  paste <(awk -F'|' 'NR>3{print $1 "|" $3 "|" $5}' $O/micro_c${ch}_prio.txt) <(awk -F'|' 'NR>3{print $3 "|" $5}' $O/micro_c${ch}_base.txt) | grep -v "^(the" | sed 's/([^)]*)//g'
done
X="--gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-multilingual --no-bf16x3-line --no-streaming-line --no-rccl-probe --no-bracket-ab --no-latency-pass --no-soak"
run() { tag=$1; shift; env "$@" timeout 600 python bench.py $X > $O/b_$tag.json 2> $O/b_$tag.err; python -c "import json; d=json.load(open('$O/b_$tag.json')); print('$tag:', d['value'], d['ms_per_step'])" || tail -3 $O/b_$tag.err; }
run p1 SS_HIP_LIB=tools/bin/libss_prio.so
run b1 A=1
run p2 SS_HIP_LIB=tools/bin/libss_prio.so
run b2 A=1
