#!/bin/bash
# Copy the summaries of tools/jobs/r02_full_b.sh (gpurun_out/r02/full_b) into profiles/ under a build tag:
#   bash tools/collect_evidence.sh v25
set -e
cd "$(dirname "$0")/.."
T=$1; O=gpurun_out/r02/full_b; P=profiles
cp $O/bench.json $P/r02_${T}_bench.json
cp $O/bench_under_rocprof.json $P/r02_${T}_bench_under_rocprof.json
cp $O/bench_1stream_under_rocprof.json $P/r02_${T}_bench_1stream_under_rocprof.json
cp "$(ls -t $O/prof_driver/*/*_kernel_stats.csv | head -1)" $P/r02_${T}_driver_bench_kernel_stats.csv
cp "$(ls -t $O/prof_1stream/*/*_kernel_stats.csv | head -1)" $P/r02_${T}_bench_1stream_kernel_stats.csv
cp $O/pytest_gpu.log $P/r02_${T}_pytest_gpu.log
cp $O/pmc_traffic.json $P/r02_pmc_traffic.json          # the file bench.py reads `roofline.traffic` from (newest r*_pmc_traffic*.json)
cp $O/pmc_FETCH_SIZE.bench.json $P/r02_${T}_pmc_pass_bench.json
ls -la $P/r02_${T}_* $P/r02_pmc_traffic.json
