#!/bin/bash
# Copy the summaries of a tools/gpu_job.sh run (gpurun_out/<TAG>/) into profiles/ under a round prefix:
#   bash tools/collect_evidence.sh r06x r06
set -e
cd "$(dirname "$0")/.."
T=$1; R=${2:-r06}; O=gpurun_out/$T; P=profiles
cpi() { [ -f "$1" ] && cp "$1" "$2" && echo "$2"; true; }
cpi $O/bench.json $P/${R}_bench.json
cpi $O/bench.detail.json $P/${R}_bench_detail.json
cpi $O/bench_under_rocprof.detail.json $P/${R}_bench_under_rocprof.json
cpi $O/bench_1stream_under_rocprof.detail.json $P/${R}_bench_1stream_under_rocprof.json
cpi "$(ls -t $O/prof_driver/*/*_kernel_stats.csv 2>/dev/null | head -1)" $P/${R}_driver_bench_kernel_stats.csv
cpi "$(ls -t $O/prof_1stream/*/*_kernel_stats.csv 2>/dev/null | head -1)" $P/${R}_bench_1stream_kernel_stats.csv
cpi $O/pytest_gpu.log $P/${R}_pytest_gpu.log
cpi $O/bench_config_parity.log $P/${R}_bench_config_parity.log
cpi $O/pmc_traffic.json $P/${R}_pmc_traffic.json          # the file bench.py reads `roofline.traffic` from (newest r*_pmc_traffic*.json)
cpi $O/pmc_FETCH_SIZE.detail.json $P/${R}_pmc_pass_bench.json
cpi $O/share_table.md $P/${R}_share_table.md
cpi $O/roofline_table.md $P/${R}_roofline_table.md
cpi $O/accuracy_vs_float64.json $P/${R}_accuracy_vs_float64.json
cpi $O/op_accuracy.json $P/${R}_op_accuracy.json
cpi $O/stream_sweep.txt $P/${R}_stream_sweep.txt
cpi $O/pack_sweep.txt $P/${R}_pack_sweep.txt
cpi $O/trace_gaps_streams.txt $P/${R}_trace_gaps_streams.txt
cpi $O/trace_gaps_1stream.txt $P/${R}_trace_gaps_1stream.txt
