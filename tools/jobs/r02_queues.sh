cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02/queues; mkdir -p $O
run() { # name, env...
  n=$1; shift
  env "$@" timeout 300 python bench.py --steps 20 --warmup 5 --no-latency-pass --no-cpu-baseline > $O/$n.json 2> $O/$n.err
  python -c "import json; d=json.load(open('$O/$n.json')); print('$n', d['value'], d['ms_per_step'])"
}
run base A=1
run q8 GPU_MAX_HW_QUEUES=8
run q8_spare16 GPU_MAX_HW_QUEUES=8 SS_SK2_SPARE_CUS=16
run q16 GPU_MAX_HW_QUEUES=16
run q2 GPU_MAX_HW_QUEUES=2
