cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r02
for b in 32 48 64; do
  for s in 8 12; do
  timeout 600 python bench.py --steps $((640 / b)) --warmup 3 --batch $b --streams $s --no-cpu-baseline --no-latency-pass 2>gpurun_out/r02/batch_$b.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('batch $b streams $s:', d['value'], d['utterances_per_sec'], 'ms/utt', d['ms_per_utterance'], d['roofline']['kernel'], d['roofline']['achieved'])" || tail -3 gpurun_out/r02/batch_$b.err
  done
done 2>&1 | tee gpurun_out/r02/batch_sweep.txt
