# Round-5 (d): the full default bench line (driver command) with the new pack_invariance / oracle_check objects
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05/d; mkdir -p $O
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; tail -3 $O/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r05/d/bench.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d.get('near_tie_rows'))
print(json.dumps(d['pack_invariance']))
print(json.dumps(d['cpu_baseline']['oracle_check'])[:1500])
PY
