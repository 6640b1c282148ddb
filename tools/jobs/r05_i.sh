# Round-5 (i): phase cycle accounting of the Winograd slab kernels (diagnostic build)
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05/i; mkdir -p $O
SS_HIP_LIB=tools/bin/libss_cwt.so timeout 600 python tools/cw_timing.py > $O/cw_timing.txt 2>&1; cat $O/cw_timing.txt
