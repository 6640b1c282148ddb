#!/usr/bin/env python
"""CPU ablation behind VERDICT r5 #2 (test infrastructure: imports oracle/): which summation order of the K = 256 ... 2048 linears
puts float32 CTC logits how far from float64?  The oracle's F.linear is replaced by an exact emulation of an MFMA fmaf chain
(16-wide slabs, MFMA e contracts k = {e, 4 + e, 8 + e, 12 + e}) with `chains` interleaved accumulators (by e) and a flush into a
running total every `block` k; everything else (attention products, convs, LayerNorm, softmax) stays torch.
  python tests/diagnostics/chain_ablation_cpu.py [seconds=3.0]"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import streamspeech_oracle as O  # noqa: E402
from streamspeech_amd import synth  # noqa: E402
from streamspeech_amd.config import ModelConfig  # noqa: E402

_real_linear = torch.nn.functional.linear


def emulated_linear(chains, block, w2_quarters=False):
    def lin(x, w, b=None):
        if x.dtype != torch.float32:
            return _real_linear(x, w, b)
        X = x.detach().numpy().astype(np.float64).reshape(-1, x.shape[-1])
        W = w.detach().numpy().astype(np.float64)
        K = X.shape[1]

        def run(k0, k1):
            acc = [np.zeros((X.shape[0], W.shape[0]), np.float32) for _ in range(chains)]
            tot = None
            for s in range(k0 // 16, k1 // 16):
                for e in range(4):
                    c = e % chains
                    for j in range(4):
                        k = 16 * s + e + 4 * j
                        acc[c] = (acc[c].astype(np.float64) + np.outer(X[:, k], W[:, k])).astype(np.float32)
                if block and (16 * (s + 1) - k0) % block == 0 and 16 * (s + 1) < k1:
                    t = acc[0] if chains == 1 else acc[0] + acc[1]
                    tot = t if tot is None else tot + t
                    acc = [np.zeros_like(acc[0]) for _ in range(chains)]
            t = acc[0] if chains == 1 else acc[0] + acc[1]
            return t if tot is None else tot + t

        if w2_quarters and K == 2048:          # round 5's fused FFN: four waves x 512 hidden columns, then ((q0 + q3) + q2) + q1 style fixed order
            q = [run(512 * i, 512 * (i + 1)) for i in range(4)]
            out = ((q[0] + q[3]) + q[2]) + q[1]
        else:
            out = run(0, K)
        out = torch.from_numpy(out.astype(np.float32)).reshape(*x.shape[:-1], W.shape[0])
        return out if b is None else out + b
    return lin


def main():
    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
    torch.set_num_threads(4)
    cfg = ModelConfig()
    sd = synth.make_model_state_dict(0, cfg)
    fb = synth.synth_fbank(11, int(secs * 100))
    with torch.inference_mode():
        L64 = O.ctc_head(O.SD(sd, dtype=torch.float64), O.encoder_forward(O.SD(sd, dtype=torch.float64), fb, cfg), "source_unigram", cfg)[3].double()
        osd = O.SD(sd)
        res = {}
        for name, lin in (("torch (this CPU's sgemm)", _real_linear), ("1 chain, no blocks (round 5 HIP; W2 as 4 x 512)", emulated_linear(1, 0, True)),
                          ("1 chain, blocks of 256", emulated_linear(1, 256)), ("2 chains, blocks of 256", emulated_linear(2, 256)),
                          ("2 chains, blocks of 256; W2 1 chain", None)):
            if lin is None:
                l2, l1 = emulated_linear(2, 256), emulated_linear(1, 256)
                lin = lambda x, w, b=None: (l1 if x.shape[-1] == 2048 else l2)(x, w, b)   # noqa: E731
            O.F.linear = lin
            try:
                L = O.ctc_head(osd, O.encoder_forward(osd, fb, cfg), "source_unigram", cfg)[3].double()
            finally:
                O.F.linear = _real_linear
            res[name] = float(((L - L64) ** 2).mean().sqrt())
            print(name, res[name], file=sys.stderr, flush=True)
    base = res["torch (this CPU's sgemm)"]
    print(json.dumps({"what": "RMS(float32 ASR CTC logits - float64), encoder linears summed as named", "seconds": secs, "rows": int(L64.shape[0]),
                      "rms": res, "ratio_to_torch": {k: round(v / base, 3) for k, v in res.items()}}, indent=1))


if __name__ == "__main__":
    main()
