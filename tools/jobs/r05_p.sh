# Round-5 (p): build() + smoke() in ONE process on the GPU box (the load-order fix), and smoke() alone
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05/p; mkdir -p $O
timeout 600 python __graft_entry__.py smoke > $O/build_then_smoke.log 2>&1; tail -2 $O/build_then_smoke.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 300 python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/build_smoke_c.log 2>&1; tail -1 $O/build_smoke_c.log
