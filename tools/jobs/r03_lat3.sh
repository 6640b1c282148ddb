cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r03
for g in 4 2 1 0.5; do for u in 6144 1024; do echo "== SS_SK_MIN_GFLOP=$g SS_SK_MIN_UNITS=$u"; SS_SK_MIN_GFLOP=$g SS_SK_MIN_UNITS=$u python tools/latency_breakdown.py 2>&1 | grep -v amdgpu.ids | grep "utterance\|vocoder\|t2u\|sum of"; done; done | tee gpurun_out/r03/latency_sk_threshold.txt
