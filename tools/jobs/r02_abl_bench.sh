cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r02
for lib in default tools/libss_k2abl8.so tools/libss_k2abl64.so tools/libss_k2abl128.so tools/libss_k2abl192.so; do
  if [ $lib = default ]; then unset SS_HIP_LIB; else export SS_HIP_LIB=$PWD/$lib; fi
  timeout 300 python bench.py --steps 20 --warmup 3 --streams 1 --no-latency-pass --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$lib', 'ms/step', d['ms_per_step'], 'RTFx', d['value'], '|', r['kernel'], 'TF', r['achieved'], 'avg us', r['avg_launch_us'], 'kernel_time/wall', r['kernel_time_over_wall'])"
done 2>&1 | tee gpurun_out/r02/abl_bench.txt
