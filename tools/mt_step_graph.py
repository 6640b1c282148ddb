"""The MT greedy decode step (ss_mt_append with one token: embed + 4 decoder layers + LN + vocab projection + argmax, 36
launches) issued eagerly vs replayed as a captured hipGraph, at a fixed position (timing only: a replay recomputes the
same step).  Answers "is the decode chain host-bound / would a graph shorten it?" with numbers.
Run on the GPU box: python tools/mt_step_graph.py"""
import ctypes as C
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from streamspeech_amd import lib as L, synth  # noqa: E402
from streamspeech_amd.config import ModelConfig  # noqa: E402
from streamspeech_amd.engine import HipModel  # noqa: E402


def main():
    hip = C.CDLL("libamdhip64.so")
    cfg = ModelConfig()
    m = HipModel(synth.make_model_state_dict(0, cfg), cfg)
    lib = m.lib
    stream = torch.cuda.Stream()
    sp = C.c_void_p(stream.cuda_stream)
    out = {}
    with torch.cuda.stream(stream):
        for Tp, pos in ((147, 12), (375, 40)):                 # 5.9-s / 15-s utterance, early / late position
            enc = torch.from_numpy(synth.uniform(5, "g_enc", (Tp, cfg.enc_dim), -1, 1)).cuda().float().contiguous()
            m.mt_begin(enc)
            toks = [cfg.eos] + [7 + i for i in range(pos)]
            m.mt_append(toks, 0, True, False)                  # fill the KV cache up to `pos`
            tok = torch.tensor([11], dtype=torch.int32).cuda()
            feats = torch.empty((1, cfg.dec_dim), device="cuda")
            nxt = torch.empty((1,), dtype=torch.int32, device="cuda")
            P = lambda t: C.c_void_p(t.data_ptr())

            def step():
                rc = lib.ss_mt_append(m.h, sp, P(tok), 1, pos + 1, 1, 0, P(feats), P(nxt), 0)
                assert rc == 0, rc
                lib.ss_mt_truncate(m.h, pos + 1)

            for _ in range(5):
                step()
            stream.synchronize()
            N = 200
            t0 = time.perf_counter()
            for _ in range(N):
                step()
            t_issue = time.perf_counter() - t0                 # host time to enqueue
            stream.synchronize()
            t_eager = time.perf_counter() - t0
            # capture one step
            graph, gexec = C.c_void_p(), C.c_void_p()
            assert hip.hipStreamBeginCapture(sp, 0) == 0
            step()
            assert hip.hipStreamEndCapture(sp, C.byref(graph)) == 0
            n_nodes = C.c_size_t(0)
            hip.hipGraphGetNodes(graph, None, C.byref(n_nodes))
            assert hip.hipGraphInstantiate(C.byref(gexec), graph, None, None, 0) == 0
            for _ in range(5):
                assert hip.hipGraphLaunch(gexec, sp) == 0
            stream.synchronize()
            t0 = time.perf_counter()
            for _ in range(N):
                hip.hipGraphLaunch(gexec, sp)
            g_issue = time.perf_counter() - t0
            stream.synchronize()
            t_graph = time.perf_counter() - t0
            hip.hipGraphExecDestroy(gexec); hip.hipGraphDestroy(graph)
            out[f"Tp{Tp}_pos{pos}"] = {"graph_nodes": int(n_nodes.value), "eager_us_per_step": round(1e6 * t_eager / N, 1),
                                       "eager_host_issue_us_per_step": round(1e6 * t_issue / N, 1),
                                       "graph_us_per_step": round(1e6 * t_graph / N, 1),
                                       "graph_host_issue_us_per_step": round(1e6 * g_issue / N, 1)}
    print(json.dumps(out, indent=1))
    os.makedirs("gpurun_out/r02", exist_ok=True)
    json.dump(out, open("gpurun_out/r02/mt_step_graph.json", "w"), indent=1)


if __name__ == "__main__":
    main()
