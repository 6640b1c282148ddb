# Round-4 (w): the whole GPU suite of the committed build, on its own (the combined final job lost its box once, cause unknown)
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04/w; mkdir -p $O
( time timeout 900 python -m pytest tests -m gpu -q ) > $O/pytest_gpu.log 2>&1; tail -14 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $O/smoke.log
