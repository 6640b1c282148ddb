# Round-5: what one CU can stream (tools/src/cu_stream_rate.hip) -- the bound of a "one launch per decoder layer" MT decode step whose
# workgroups own whole rows
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r05
( for nt in 256 1024; do ./tools/bin/cu_stream_rate 8 $nt; done; ./tools/bin/cu_stream_rate 12 1024; ./tools/bin/cu_stream_rate 2 1024 ) 2>&1 | tee gpurun_out/r05/cu_stream_rate.txt
