# The N > 1 code path of bench.py on the one GPU a lease has: two ranks under torch.distributed.run, RCCL backend, both on
# device 0 (bench.py takes LOCAL_RANK modulo the device count).  Not a scaling measurement -- a readiness artefact.
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r02
export HSA_ENABLE_IPC_MODE_LEGACY=0
for be in nccl gloo; do
  SS_DIST_BACKEND=$be timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 6 --warmup 2 --no-latency-pass > gpurun_out/r02/two_ranks_$be.json 2> gpurun_out/r02/two_ranks_$be.err
  echo "backend $be rc=$?"; tail -c 600 gpurun_out/r02/two_ranks_$be.json; tail -3 gpurun_out/r02/two_ranks_$be.err
done
