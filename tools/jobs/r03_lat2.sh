cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r03
for m in 128 192 256 384; do echo "== SS_SMALLM_MAX_ROWS=$m"; SS_SMALLM_MAX_ROWS=$m python tools/latency_breakdown.py 2>&1 | grep -v amdgpu.ids | grep "utterance\|encoder\|t2u\|sum of"; done | tee gpurun_out/r03/latency_smallm_rows.txt
