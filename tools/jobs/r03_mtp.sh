cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03/mtp; mkdir -p $O
for g in 64 128 256; do
  echo "== SS_MT_PERSISTENT=$g"
  ( SS_MT_PERSISTENT=$g timeout 600 python -m pytest tests/test_stages_gpu.py tests/test_reference_agent_gpu.py -x -q -k "mt or greedy or generator or agent" ) > $O/tests_$g.log 2>&1; tail -3 $O/tests_$g.log
  SS_MT_PERSISTENT=$g timeout 300 python tools/latency_breakdown.py 2>&1 | grep -v amdgpu.ids | grep "utterance\|mt greedy\|sum of"
  SS_MT_PERSISTENT=$g timeout 120 python - <<'P'
from streamspeech_amd import lib as L
print("bounded-wait time-outs:", L.load().ss_debug_sk_errors())
P
done 2>&1 | tee $O/summary.txt
