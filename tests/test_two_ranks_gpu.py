"""Two RCCL ranks on two devices: `bench.py --gpus 2` exactly as the driver launches it -- torch.distributed.run,
one rank per GPU, backend "nccl" (= RCCL over xGMI) -- on a short workload.  Skips when the box shows fewer than two GPUs (the
builder's lease has one; the driver's boxes may have more).  What the N > 1 path shards, which collectives it uses and how the
line is formed is otherwise covered on CPU over gloo (tests/test_dp_gloo.py, tests/test_bench_line_cpu.py).
Reference shape: fairseq/fairseq/distributed/utils.py:595-666 (all_gather_list merge of sharded generation)."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_bench_two_rccl_ranks_on_two_devices(tmp_path):
    if not torch.cuda.is_available() or torch.cuda.device_count() < 2:
        pytest.skip(f"needs two GPUs, this box shows {torch.cuda.device_count() if torch.cuda.is_available() else 0}")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", SS_BENCH_DETAIL=str(tmp_path / "detail.json"))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--batch", "32",
           "--streams", "2"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and len(lines[0]) < 6000, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["value"] > 0 and line["scaling"] == "weak"
    assert [p["rank"] for p in line["per_rank"]] == [0, 1] and all(p["utterances"] == 4 * 32 for p in line["per_rank"])
    assert line["rccl"]["backend"] == "nccl" and line["rccl"]["world"] == 2 and line["rccl"]["results_ok"] is True
    assert line["cpu_baseline"] is None and "soak" not in line and "multilingual" not in line      # N > 1: no optional legs
    # value = units of all ranks / MAX wall over ranks
    wall = max(p["wall_s"] for p in line["per_rank"])
    audio = sum(p["audio_s"] for p in line["per_rank"])
    assert abs(line["value"] - audio / wall) / line["value"] < 1e-3
