cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02/trace; mkdir -p $O
export TMPDIR=/tmp
for st in 8 1; do
SS_BENCH_NO_REPLAY=1 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$st -- python bench.py --steps 40 --warmup 5 --streams $st --no-latency-pass --no-cpu-baseline > $O/b_$st.json 2> $O/b_$st.err
F=$(ls /tmp/tr_$st/*/*kernel_trace.csv | head -1)
head -2 $F > $O/trace_head_$st.txt
python tools/trace_gaps.py $F $O/b_$st.json | tee $O/gaps_$st.txt
done
