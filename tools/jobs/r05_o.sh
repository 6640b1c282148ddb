# Round-5 (o): phase skew between workgroups of the Winograd slab kernels (SS_CW_SKEW_US): micro at 64 / 128 / 256 channels, bench
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05/o; mkdir -p $O
for ch in 64 128 256; do
  for sk in 0 4 8 16; do
    SS_CW_SKEW_US=$sk C64_BENCH_CHANNELS=$ch timeout 300 python tools/c64_bench.py > $O/micro_c${ch}_skew$sk.txt 2>&1
  done
  echo "== $ch channels: slab conv1 | conv2 at skew 0 / 4 / 8 / 16 us"
  paste <(awk -F'|' 'NR>3{print $1 "|" $3 "|" $5}' $O/micro_c${ch}_skew0.txt) <(awk -F'|' 'NR>3{print $3 "|" $5}' $O/micro_c${ch}_skew4.txt) <(awk -F'|' 'NR>3{print $3 "|" $5}' $O/micro_c${ch}_skew8.txt) <(awk -F'|' 'NR>3{print $3 "|" $5}' $O/micro_c${ch}_skew16.txt) | grep -v "^(the" | sed 's/([^)]*)//g'
done
X="--gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-multilingual --no-bf16x3-line --no-streaming-line --no-rccl-probe --no-bracket-ab --no-latency-pass --no-soak"
run() { tag=$1; shift; env "$@" timeout 600 python bench.py $X > $O/b_$tag.json 2> $O/b_$tag.err; python -c "import json; d=json.load(open('$O/b_$tag.json')); print('$tag:', d['value'], d['ms_per_step'])" || tail -3 $O/b_$tag.err; }
run s0a SS_CW_SKEW_US=0
run s8a SS_CW_SKEW_US=8
run s0b SS_CW_SKEW_US=0
run s8b SS_CW_SKEW_US=8
run s16 SS_CW_SKEW_US=16
run s4 SS_CW_SKEW_US=4
