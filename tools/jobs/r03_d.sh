cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03/d; mkdir -p $O
( timeout 900 python -m pytest tests -m gpu -x -q -k "mt or agent or decode or attention or stages or batch_mt" ) > $O/tests.log 2>&1; tail -4 $O/tests.log
python tools/latency_breakdown.py 2>&1 | grep -v amdgpu.ids | tee $O/latency_breakdown.txt | grep "utterance\|mt greedy\|t2u\|sum of"
