#!/bin/bash
# round 5, job a: first run of the pack-invariant routes -- new tests + the suites closest to the change
mkdir -p gpurun_out/r05
python -m pytest tests/test_pack_invariance_gpu.py -x -q -m gpu > gpurun_out/r05/a_pack_invariance.log 2>&1
tail -30 gpurun_out/r05/a_pack_invariance.log
python -m pytest tests/test_ffn_gpu.py tests/test_rtlin_gpu.py tests/test_margin_gpu.py tests/test_batch_gpu.py -q -m gpu > gpurun_out/r05/a_near.log 2>&1
tail -30 gpurun_out/r05/a_near.log
