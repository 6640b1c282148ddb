cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r03
for pw in 0 1 0 1; do echo "== SS_SK2_PW=$pw"; SS_SK2_PW=$pw SK2_SHAPES="stage2 k11,stage2 k7,stage2 k3,up3" timeout 120 python tools/sk2_bench.py 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r03/sk2_pw.txt
