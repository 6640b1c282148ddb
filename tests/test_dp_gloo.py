"""N > 1 path on CPU: world_size-2 gloo run of the utterance-DP sharding and statistics reduction
that bench.py uses (no GPU, no HIP compute)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from streamspeech_amd import dp, workload


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]



def _run_cli(cmd, env, timeout):
    """subprocess.run(capture_output=True, timeout=...) that cannot hang: the command runs in its own process group, and on a
    time-out the WHOLE group is killed (a launcher killed alone leaves its ranks holding the pipes, and communicate() then waits
    for ever)."""
    import signal
    import subprocess
    p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True)
    try:
        out, err = p.communicate(timeout=timeout)
    except subprocess.TimeoutExpired:
        os.killpg(p.pid, signal.SIGKILL)
        out, err = p.communicate()
        raise AssertionError(f"timed out after {timeout} s: {' '.join(cmd)}\n{err[-2000:]}")
    return subprocess.CompletedProcess(cmd, p.returncode, out, err)


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    utts = workload.make_utterances(10)
    mine = dp.shard(utts, rank, world)
    dist.barrier()
    wall = 1.0 + rank            # rank 1 is the straggler
    audio = sum(u.seconds for u in mine)
    pr = dp.gather_per_rank(dist, wall, audio, float(len(mine)))
    w, a, n = dp.reduce_stats(dist, wall, audio, float(len(mine)))
    assert [x["rank"] for x in pr] == [0, 1] and [x["wall_s"] for x in pr] == [1.0, 2.0]
    assert abs(pr[rank]["audio_s"] - round(audio, 2)) < 1e-9 and pr[rank]["utterances"] == len(mine)
    q.put((rank, [u.idx for u in mine], w, a, n))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_sharding_and_reduction():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    utts = workload.make_utterances(10)
    idx = sorted(res[0][1] + res[1][1])
    assert idx == list(range(10)) and not set(res[0][1]) & set(res[1][1])      # a partition
    total_audio = sum(u.seconds for u in utts)
    for _, _, w, a, n in res:
        assert w == 2.0 and abs(a - total_audio) < 1e-9 and n == 10.0          # MAX wall, SUM audio / utterances


def test_balanced_shards():
    durs = [u.seconds for u in workload.make_utterances(64)]
    sh = dp.balanced_shards(durs, 8)
    assert sorted(i for s in sh for i in s) == list(range(64))
    loads = [sum(durs[i] for i in s) for s in sh]
    assert max(loads) / min(loads) < 1.15


def test_strong_scaling_plan_partitions_one_fixed_set():
    """bench.py --scaling strong (BASELINE.json configs[3] literally: ONE 1024-utterance set split over the GPUs):
    the ranks' timed utterances partition the N = 1 set, audio seconds balance, and the weak plan is unchanged."""
    steps, batch = 32, 32
    one, g1 = workload.bench_plan(steps, batch, 0, 1, strong=True)
    ref = sorted((one[i].idx, one[i].n_samples) for g in g1 for i in g)
    assert len(ref) == steps * batch and len(g1) == steps
    for world in (2, 4, 8):
        seen, audio = [], []
        for r in range(world):
            mine, groups = workload.bench_plan(steps, batch, r, world, strong=True)
            assert len(groups) == steps // world and all(len(g) == batch for g in groups)
            ids = [(mine[i].idx, mine[i].n_samples) for g in groups for i in g]
            seen += ids
            audio.append(sum(mine[i].seconds for g in groups for i in g))
            for g in groups:        # length-bucketed: batches hold neighbours of the sorted order
                secs = [mine[i].seconds for i in g]
                assert secs == sorted(secs, reverse=True)
        assert sorted(seen) == ref                                              # a partition of the very set N = 1 runs
        assert max(audio) / min(audio) < 1.02
    # fewer steps than ranks: the 64 utterances still split 8 ways (one short batch per rank)
    assert sum(len(workload.bench_plan(2, 32, r, 8, strong=True)[1]) for r in range(8)) == 8 and \
        all(len(g) == 8 for r in range(8) for g in workload.bench_plan(2, 32, r, 8, strong=True)[1])


def test_bench_gpus_2_as_typed_self_launches_its_ranks():
    """`python bench.py --gpus 2 ...` typed WITHOUT a launcher must start its own ranks (it used to
    SystemExit).  --dry-plan keeps it on CPU: plan per rank + gloo barrier / MAX / SUM / all-gather, rank 0 prints the
    one JSON line, every other rank nothing, exit code 0."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = _run_cli([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--batch", "64", "--dry-plan"], env, 240)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = r.stdout.splitlines()            # the bench line and NOTHING else on stdout (library banners go to stderr)
    assert len(lines) == 1 and lines[0].startswith("{"), r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["self_launched"] and out["dry_plan"]
    assert [p["rank"] for p in out["per_rank"]] == [0, 1] and all(p["utterances"] == 4 * 64 for p in out["per_rank"])
    assert out["stand_in_wall_max_s"] == 1.001                    # MAX over ranks picked rank 1's
    assert out["planned_utterances"] == 512 and out["rccl"]["results_ok"] and out["rccl"]["world"] == 2   # the collectives' summary (gloo here)
    assert len(lines[0]) < 6000
    # strong scaling: the one fixed set splits over the ranks
    r = _run_cli([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--batch", "64", "--dry-plan", "--scaling", "strong"], env, 240)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert out["planned_utterances"] == 256 and out["steps_per_gpu"] == 2


def test_bench_failing_rank_gives_nonzero_exit():
    """A rank that dies must fail the whole command (the launcher's exit code is bench.py's)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["SS_BENCH_DRY_FAIL_RANK"] = "1"
    r = _run_cli([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--dry-plan"], env, 240)
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_bench_gpus_8_dry_plan_balances_audio_and_prints_one_line():
    """the driver's SCALE command shape at N = 8 on CPU -- `bench.py --gpus 8 --steps 20 --warmup 5 --dry-plan` as
    typed: eight ranks over gloo, every rank's planned audio seconds within 1 % of the mean, exactly one JSON line."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    r = _run_cli([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "20", "--warmup", "5", "--dry-plan"], env, 300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = r.stdout.splitlines()
    assert len(lines) == 1 and lines[0].startswith("{"), r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["dry_plan"] and out["scaling"] == "weak"
    audio = [p["audio_s"] for p in out["per_rank"]]
    assert [p["rank"] for p in out["per_rank"]] == list(range(8)) and all(p["utterances"] == 20 * 128 for p in out["per_rank"])
    mean = sum(audio) / 8
    assert max(abs(a - mean) for a in audio) / mean < 0.01, audio
    assert out["rccl"]["results_ok"] and out["rccl"]["world"] == 8 and len(lines[0]) < 6000
    # (per-rank host placement and the collectives' timings are sidecar material: tests/test_bench_line_cpu.py reads them there)


def test_numa_pinning_picks_the_gpus_local_cpus(tmp_path):
    """dp.pin_rank_to_gpu_numa against a fake sysfs: a device on node 1 whose local CPUs are a proper subset of the allowed ones."""
    from streamspeech_amd import dp
    allowed = sorted(os.sched_getaffinity(0))
    if len(allowed) < 2:
        import pytest
        pytest.skip("one CPU")
    local = allowed[: max(1, len(allowed) // 2)]
    d = tmp_path / "0000:c1:00.0"
    d.mkdir()
    (d / "numa_node").write_text("1\n")
    (d / "local_cpulist").write_text(",".join(str(c) for c in local) + "\n")
    assert dp.gpu_local_cpus("0000:C1:00.0", str(tmp_path)) == (1, local)
    info = dp.pin_rank_to_gpu_numa(0, 8, min_cpus=1, pci_bus_id="0000:c1:00.0", sysfs_root=str(tmp_path), apply=False)
    assert info["numa_node"] == 1 and info["cpus_local_to_gpu"] == len(local) and info["pinned"] is False   # apply=False: decision only
    one = dp.pin_rank_to_gpu_numa(0, 1, min_cpus=1, pci_bus_id="0000:c1:00.0", sysfs_root=str(tmp_path))
    assert one["pinned"] is False and "one rank" in one["why"]
    import subprocess
    import sys
    code = ("import os,sys; sys.path.insert(0, %r); from streamspeech_amd import dp; "
            "i = dp.pin_rank_to_gpu_numa(0, 8, min_cpus=1, pci_bus_id='0000:c1:00.0', sysfs_root=%r); "
            "print(i['pinned'], sorted(os.sched_getaffinity(0)) == %r)") % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), str(tmp_path), local)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.stdout.split() == ["True", "True"], (r.stdout, r.stderr[-500:])
    (d / "numa_node").write_text("-1\n")
    assert dp.pin_rank_to_gpu_numa(0, 8, pci_bus_id="0000:c1:00.0", sysfs_root=str(tmp_path))["pinned"] is False
