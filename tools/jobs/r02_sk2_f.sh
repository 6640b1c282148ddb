cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r02
timeout 600 python -m pytest tests/test_ops_gpu.py -q -k "stream_k" 2>&1 | tail -3
SK2_SHAPES="stage0 k11,stage0 k7,stage0 k3,stage1 k11,stage1 k7,stage1 k3,up1,up2,up0,conv_pre,unit fc1,unit fc2,unit qkv,enc ffn1" timeout 900 python tools/sk2_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02/sk2_bench_f.txt
bash tools/jobs/r02_bench_1s.sh 2>&1 | tail -4
