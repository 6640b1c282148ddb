# Round-4 (x): further counters of the one-stream form (own pass each, --kernel-trace only): LDS bank conflicts, VALU / memory-unit busy,
# occupancy -- what the MFMA kernels' non-MFMA cycles are made of.  Then the 200-step soak at 64 utterances per batch.
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04/x; mkdir -p $O
export TMPDIR=/tmp
rocprofv3 -L > $O/avail.txt 2>&1
X="--steps 6 --warmup 1 --streams 1 --no-latency-pass --no-cpu-baseline --no-bf16x3-line --no-multilingual --no-streaming-line --no-bracket-ab --no-rccl-probe"
SPECS=""
for c in LdsBankConflict VALUBusy MemUnitStalled MemUnitBusy OccupancyPercent SALUBusy; do
  if grep -q "$c" $O/avail.txt; then
    timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -- python bench.py $X > $O/pmc_$c.bench.json 2> $O/pmc_$c.err
    F=$(ls $O/pmc_$c/*/*counter_collection.csv 2>/dev/null | head -1)
    [ -n "$F" ] && SPECS="$SPECS $c=$F"
  else
    echo "counter $c not offered by rocprofv3 -L on this box"
  fi
done
python tools/pmc_extra.py $O/pmc_extra.md $SPECS
rm -f $O/pmc_*/*/*counter_collection.csv $O/pmc_*/*/*kernel_trace.csv
grep -i -A3 "LdsBankConflict\|MemUnitStalled" $O/avail.txt | head -40 > $O/avail_excerpt.txt; rm -f $O/avail.txt
timeout 600 python bench.py --steps 200 --warmup 10 --no-latency-pass --no-cpu-baseline --no-bf16x3-line --no-multilingual --no-streaming-line --no-bracket-ab --no-rccl-probe > $O/soak_200_steps.json 2> $O/soak.err; python -c "import json; d=json.load(open('$O/soak_200_steps.json')); print('soak 200 steps:', d['value'], d['ms_per_step'], d['roofline']['frac'], d.get('stream_k_spin_timeouts'))"
