# Round-5 (j): staging loads requested before the block-start barrier (CW_EARLY_LOAD): micro, phase accounting, bench A/B against a build without
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05/j; mkdir -p $O
for ch in 128 64; do
  C64_BENCH_CHANNELS=$ch timeout 300 python tools/c64_bench.py > $O/micro_c${ch}_early.txt 2>&1
  SS_HIP_LIB=tools/bin/libss_cwl0.so C64_BENCH_CHANNELS=$ch timeout 300 python tools/c64_bench.py > $O/micro_c${ch}_late.txt 2>&1
  paste <(awk -F'|' 'NR>3{print $1 "|" $3 "|" $5}' $O/micro_c${ch}_early.txt) <(awk -F'|' 'NR>3{print $3 "|" $5}' $O/micro_c${ch}_late.txt)
done
SS_HIP_LIB=tools/bin/libss_cwt.so timeout 600 python tools/cw_timing.py > $O/cw_timing_early.txt 2>&1; grep -v "^256" $O/cw_timing_early.txt | cut -c1-110
X="--gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-multilingual --no-bf16x3-line --no-streaming-line --no-rccl-probe --no-bracket-ab --no-latency-pass --no-soak"
run() { tag=$1; shift; env "$@" timeout 600 python bench.py $X > $O/b_$tag.json 2> $O/b_$tag.err; python -c "import json; d=json.load(open('$O/b_$tag.json')); print('$tag:', d['value'], d['ms_per_step'])" || tail -3 $O/b_$tag.err; }
run e1 A=1
run l1 SS_HIP_LIB=tools/bin/libss_cwl0.so
run e2 A=1
run l2 SS_HIP_LIB=tools/bin/libss_cwl0.so
run e3 A=1
run l3 SS_HIP_LIB=tools/bin/libss_cwl0.so
