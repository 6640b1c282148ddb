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
