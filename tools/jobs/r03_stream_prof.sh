# Kernel statistics of the streaming configuration (configs[2]): the 320-ms agent policy() loop with the persistent MT
# decode step (default) and with the launch-per-op step (--mt-step-workgroups 0 equivalent: SS_BENCH_MT_WGS=0).
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03/stream; mkdir -p $O
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python bench.py --mode streaming --segment-ms 320 --utterances 12 --no-cpu-baseline > $O/stream_bench.json 2> $O/prof.err
S=$(ls -t $O/prof/*/*_kernel_stats.csv | head -1); cp $S $O/streaming_kernel_stats.csv
head -14 $O/streaming_kernel_stats.csv | cut -c1-200
python -c "
import json; d=json.load(open('$O/stream_bench.json'))
for k in ('incremental','full_recompute','incremental_launch_per_op_mt'):
    c=d.get(k) or {}; print(k, c.get('rtfx_compute'), c.get('ms_per_policy_call_mean'), c.get('ms_per_policy_call_p95'), c.get('gemm_class_launches_per_policy_call'))"
rm -f $O/prof/*/*kernel_trace.csv
