# Round-5 (r): the driver's bench command once more with profiles/r05_pmc_traffic.json in place (roofline.traffic joins to it), the new
# duration-invariance test and the suites around the vocoder
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05/r; mkdir -p $O
timeout 900 python -m pytest tests/test_pack_invariance_gpu.py tests/test_batch_gpu.py tests/test_stages_gpu.py -q -m gpu -k "pack or vocoder or durations or batch" > $O/tests.log 2>&1; tail -4 $O/tests.log
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
python - <<PY
import json
d = json.load(open("$O/bench.json"))
r = d["roofline"]
print(d["value"], d["ms_per_step"], r["kernel"], r["frac"], r["traffic"], r["traffic_over_algorithmic"], r["algorithmic_mbytes_per_launch_of_the_pmc_run"], r["traffic_detail"]["same_kernel_sources"], r["traffic_detail"]["files_changed_since_pmc_run"], r["traffic_detail"]["dominant_kernel_sources_unchanged"], r["traffic_detail"]["mfma_util_pct"])
print(d["soak"]["value"], d["near_tie_rows"], d["pack_invariance"]["alone_equals_in_pack_bitwise"])
PY
