cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r02
SK2_SHAPES="enc ffn1,enc ffn2,enc qkv,enc out,enc pw1,ctc head,unit out,unit head,sub conv1,short s0 k11,short s1 k3,short s0 k3,long s0 k7" timeout 900 python tools/sk2_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02/sk2_bench_g.txt
