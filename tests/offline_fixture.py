"""Shared by the CPU and GPU offline-generator tests: runs streamspeech_amd/offline.generate on the fixture's utterances
and compares its files, line by line, with what the reference's generator classes printed
(tests/golden/offline_generator.json, written by oracle/make_golden_offline.py)."""
import json
import os

import numpy as np
import torch

from oracle.make_golden_offline import sample_pcm, sample_targets

FIX = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "offline_generator.json")


def load():
    return json.load(open(FIX, encoding="utf-8"))


def ref_dicts(cfg):
    """The synthetic dictionaries the fixture was generated with (oracle/ref_agent.py make_dicts: pure Python)."""
    from oracle.ref_agent import make_dicts
    return make_dicts(cfg)


def run_and_compare(model, vocoder, cfg, group, tmp_path, device="cpu", batch_size=32, score_rel=1e-5, pos_abs=2e-4):
    from streamspeech_amd import offline
    fix = load()
    g = fix["groups"][group]
    meta = {s["id"]: s for s in fix["samples"]}
    ids = [int(i) for i in g["hypotheses"]]
    items = [(i, torch.from_numpy(sample_pcm(meta[i]["pcm_seed"], meta[i]["n_samples"])).to(device)) for i in ids]
    targets = {i: sample_targets(i) for i in ids if sample_targets(i) is not None}
    dicts = ref_dicts(cfg)
    hyps = offline.generate(model, vocoder, items, dicts, str(tmp_path), "test", batch_size=batch_size, max_len_b_mt=g["max_len_b_mt"],
                            dur_prediction=True, dump_wav=vocoder is not None, scores=True, targets=targets)
    log = open(os.path.join(tmp_path, "generate-test.log"), encoding="utf-8").read().splitlines()
    res = open(os.path.join(tmp_path, "generate-test.txt"), encoding="utf-8").read().splitlines()
    worst_pos = worst_score = 0.0
    for i in ids:
        ref = g["hypotheses"][str(i)]
        mine_log = [ln for ln in log if ln.split("\t")[0] in (f"A-{i}", f"S-{i}", f"D-{i}")]
        assert mine_log == ref["log"], f"sample {i}: A-/S-/D- lines differ"          # text lines: exact
        mine_res = [ln for ln in res if ln.split("\t")[0].split("-")[1] == str(i)]
        assert [ln.split("\t")[0] for ln in mine_res] == [ln.split("\t")[0] for ln in ref["result"]], f"sample {i}: line kinds / order"
        for a, b in zip(mine_res, ref["result"]):
            fa, fb = a.split("\t"), b.split("\t")
            if fa[0].startswith("T-"):
                assert a == b
            elif fa[0].startswith(("H-", "D-")):
                assert fa[2] == fb[2], f"sample {i}: unit string"
                rel = abs(float(fa[1]) - float(fb[1])) / max(1.0, abs(float(fb[1])))
                worst_score = max(worst_score, rel)
                assert rel < score_rel, f"sample {i}: score {fa[1]} vs {fb[1]}"
            else:                                                                     # P-
                pa, pb = np.array(fa[1].split(), np.float64), np.array(fb[1].split(), np.float64)
                assert pa.shape == pb.shape == (ref["n_positions"],)
                worst_pos = max(worst_pos, float(np.abs(pa - pb).max()))
                assert np.abs(pa - pb).max() < pos_abs, f"sample {i}: positional scores"
        assert hyps[i]["units"] == ref["units"]
    return {"worst_score_rel": worst_score, "worst_pos_abs": worst_pos, "hyps": hyps, "ids": ids}
