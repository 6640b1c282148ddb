set -x
mkdir -p gpurun_out/r02
cd "$GRAFT_REPO_ROOT"
( time python -m pytest tests -m gpu -x -q 2>&1 | tail -25 ) > gpurun_out/r02/pytest_gpu.log 2>&1
python bench.py --steps 20 --warmup 5 > gpurun_out/r02/bench_v21.json 2> gpurun_out/r02/bench_v21.err
tail -c 3000 gpurun_out/r02/bench_v21.json
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02/prof_1stream -- python bench.py --steps 20 --warmup 5 --streams 1 --no-latency-pass --no-cpu-baseline > gpurun_out/r02/bench_v21_1stream_rocprof.json 2> gpurun_out/r02/prof_1stream.err
ls -R gpurun_out/r02/prof_1stream | head -20
cat gpurun_out/r02/pytest_gpu.log | tail -30
