cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02/spare; mkdir -p $O
for s in 0 8 16 32; do
  SS_SK2_SPARE_CUS=$s timeout 300 python bench.py --steps 20 --warmup 5 --no-latency-pass --no-cpu-baseline > $O/b_$s.json 2> $O/b_$s.err
  python -c "import json; d=json.load(open('$O/b_$s.json')); print('spare', $s, d['value'], d['ms_per_step'])"
done
for st in 4 12 16; do
  timeout 300 python bench.py --steps 20 --warmup 5 --streams $st --no-latency-pass --no-cpu-baseline > $O/s_$st.json 2> $O/s_$st.err
  python -c "import json; d=json.load(open('$O/s_$st.json')); print('streams', $st, d['value'], d['ms_per_step'])"
done
