"""Offline batch driver with the reference's `fairseq-generate` output format.

Reference flow (researches/ctc_unity/test_scripts/pred.offline-s2st.sh): `fairseq-generate --task
speech_to_speech_ctc ...` prints, per utterance, `A-<id>\\t<asr text>`, `S-<id>\\t<ctc target text>` and
`D-<id>\\t<translation>` into generate-<subset>.log (sequence_generator_multi_decoder_ctc.py:227,248,289)
and `H-/D-/P-<id>` lines with the unit sequence into <results-path>/generate-<subset>.txt
(fairseq_cli/generate.py:257-300); the script then cuts `.asr/.tgt/.unit` files out of those and
runs examples/speech_to_speech/generate_waveform_from_code.py, which writes `<line-no>_pred.wav`.
This driver produces the same files from the HIP path, so the reference's scoring scripts
(sacrebleu / wer / asr_bleu) run unchanged on its output.

Batching mirrors fairseq: utterances are ordered by length (dataset.ordered_indices) and cut into
batches; each batch runs as one ragged no-padding pack through the ss_batch_* entry points (every
utterance keeps its B = 1 arithmetic).  Sharding mirrors `--num-shards/--shard-id`.

The `score` column of H-/D- is the sum of the per-position maximum log-probabilities in the
reference (ctc_generator.py:60-91); the HIP path takes the argmax of the logits without ever forming
log-probabilities, so the column is written as 0 and `P-` lines are omitted unless --scores is
given (then both come from ss_row_max_logprob over the unit logits of the single-utterance entry
point: one more kernel, no torch arithmetic).  `T-` lines (generate.py:258-259) are written when
the manifest carries target units (`tgt_audio` column, as the reference's S2UT manifests do).

Pinned against the reference's own generator classes run on CPU (oracle/ref_offline.py ->
tests/golden/offline_generator.json; tests/test_offline_generator_cpu.py, tests/test_offline_generator_gpu.py).
"""
import argparse
import math
import os
import sys
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import frontend
from .pipeline import units_from_tokens


def detok(symbols: Sequence[str]) -> str:
    """sequence_generator_multi_decoder_ctc.py:218-226: join, '_' and the SentencePiece mark become
    spaces, <unk> a space, <s>/</s> vanish, one leading space is dropped."""
    text = "".join(symbols)
    for a, b in (("_", " "), ("▁", " "), ("<unk>", " "), ("<s>", ""), ("</s>", "")):
        text = text.replace(a, b)
    return text[1:] if text.startswith(" ") else text


def ordered_batches(lengths: Sequence[int], batch_size: int, max_tokens: int = 0) -> List[List[int]]:
    """Length-sorted batches (longest first) bounded by `batch_size` utterances and, if > 0, by
    `max_tokens` = batch_count * longest_length (fairseq batch_by_size semantics for speech input)."""
    order = sorted(range(len(lengths)), key=lambda i: (-lengths[i], i))
    out, cur = [], []
    for i in order:
        longest = lengths[cur[0]] if cur else lengths[i]
        if cur and (len(cur) >= batch_size or (max_tokens > 0 and (len(cur) + 1) * longest > max_tokens)):
            out.append(cur)
            cur = []
        cur.append(i)
    if cur:
        out.append(cur)
    return out


def generate(model, vocoder, items: Sequence[Tuple[int, torch.Tensor]], dicts: Dict[str, object], results_path: str,
             subset: str = "test", batch_size: int = 32, max_tokens: int = 0, max_len_a: float = 0.0,
             max_len_b: int = 200, max_len_a_mt: float = 0.0, max_len_b_mt: int = 200, dur_prediction: bool = True, dump_wav: bool = True, t2u_causal: bool = False,
             scores: bool = False, log=None, targets: Optional[Dict[int, Sequence[int]]] = None) -> Dict[int, Dict]:
    """items: (sample id, 16 kHz float PCM in [-1, 1] on the device).  Writes generate-<subset>.log/.txt,
    the cut .asr/.tgt/.unit files and pred_wav/<n>_pred.wav; returns the per-id hypotheses."""
    cfg = model.cfg
    os.makedirs(results_path, exist_ok=True)
    log_f = log or open(os.path.join(results_path, f"generate-{subset}.log"), "w", encoding="utf-8")
    res_f = open(os.path.join(results_path, f"generate-{subset}.txt"), "w", encoding="utf-8")
    hyps: Dict[int, Dict] = {}
    lens = [int(p.numel()) for _, p in items]
    for sid, pcm in items:                      # shorter than one 25-ms fbank window: nothing to decode
        if pcm.numel() < 400:
            for tag in "ASD":
                print(f"{tag}-{sid}\t", file=log_f)
            print(f"H-{sid}\t0.0\t", file=res_f)
            print(f"D-{sid}\t0.0\t", file=res_f)
            hyps[sid] = {"asr": "", "st": "", "mt": "", "units": [], "wav": None}
    keep = [i for i in range(len(items)) if lens[i] >= 400]
    for group_k in ordered_batches([lens[i] for i in keep], batch_size, max_tokens):
        group = [keep[j] for j in group_k]
        ids = [items[i][0] for i in group]
        pcm = torch.cat([items[i][1].reshape(-1) for i in group])
        feat, T = model.batch_fbank_cmvn(pcm, [lens[i] for i in group])
        enc, Tp = model.batch_encoder_forward(feat, T)
        asr = model.batch_ctc_greedy(0, enc, Tp)
        st = model.batch_ctc_greedy(1, enc, Tp)
        # first-pass text search: max_len = min(int(max_len_a_mt * src_len + max_len_b_mt), max positions - 1) with
        # the task's defaults 0 / 200 (tasks/speech_to_speech_ctc.py:39-40 -> sequence_generator_multi_decoder_ctc.py
        # :130-131); src_len is the fbank frame count.  --max-len-a/-b configure the (NAR) unit generator, which has
        # no length search here.
        mx = [min(int(max_len_a_mt * t + max_len_b_mt), cfg.max_target_positions - 1) for t in T]
        toks, feats, n = model.batch_mt_greedy(enc, Tp, mx)
        unit_toks = model.batch_t2u_units(feats, n, t2u_causal=t2u_causal, mask_eos=True)
        codes = [units_from_tokens(t, cfg) for t in unit_toks]
        have = [b for b, c in enumerate(codes) if len(c) > 0]
        wavs: Dict[int, torch.Tensor] = {}
        if dump_wav and have:
            w, _, _ = vocoder.batch_forward([codes[b] for b in have], dur_prediction=dur_prediction)
            wavs = {b: w[j] for j, b in enumerate(have)}
        for b, sid in enumerate(ids):
            a_txt = detok([dicts["source_unigram"][c] for c in asr[b][0]])
            s_txt = detok([dicts["ctc_target_unigram"][c] for c in st[b][0]])
            mt = [t for t in toks[b] if t != cfg.eos]
            d_txt = detok([dicts["target_unigram"][c] for c in mt])
            print(f"A-{sid}\t{a_txt}", file=log_f)
            print(f"S-{sid}\t{s_txt}", file=log_f)
            print(f"D-{sid}\t{d_txt}", file=log_f)
            unit_str = " ".join(str(u) for u in codes[b])
            score, pos = 0.0, None
            if scores:
                score, pos = _unit_scores(model, feats[b][: n[b]], t2u_causal)
            if targets is not None and sid in targets:
                print(f"T-{sid}\t" + " ".join(str(u) for u in targets[sid]), file=res_f)
            print(f"H-{sid}\t{score}\t{unit_str}", file=res_f)
            print(f"D-{sid}\t{score}\t{unit_str}", file=res_f)
            if pos is not None:
                print(f"P-{sid}\t" + " ".join("{:.4f}".format(x) for x in pos), file=res_f)
            hyps[sid] = {"asr": a_txt, "st": s_txt, "mt": d_txt, "units": codes[b], "wav": wavs.get(b)}
    res_f.close()
    if log is None:
        log_f.close()
    _cut_files(hyps, results_path, subset, dump_wav)
    return hyps


def _unit_scores(model, feats: torch.Tensor, t2u_causal: bool):
    """Sum / per-position max log-probabilities in base 2 (generate.py:274,289), pad / unk / eos masked as in
    ctc_generator.py:55-59.  The log-softmax + max runs in the engine (HIP: ss_row_max_logprob); the base change and the
    sum (a float64 host sum like `scores[b].sum()` -> utils.item) are glue."""
    best = model.unit_scores(feats.contiguous(), t2u_causal=t2u_causal).double() / math.log(2)
    return float(best.sum()), best.tolist()


def _cut_files(hyps: Dict[int, Dict], results_path: str, subset: str, dump_wav: bool):
    """What pred.offline-s2st.sh greps/sorts/cuts out of the two generate files, and the wav dump of
    generate_waveform_from_code.py (file name = line number in the sorted .unit file)."""
    ids = sorted(hyps)
    for ext, key in ((".asr", "asr"), (".tgt", "mt")):
        with open(os.path.join(results_path, f"generate-{subset}{ext}"), "w", encoding="utf-8") as f:
            for i in ids:
                print(hyps[i][key], file=f)
    with open(os.path.join(results_path, f"generate-{subset}.unit"), "w", encoding="utf-8") as f:
        for i in ids:
            print(" ".join(str(u) for u in hyps[i]["units"]), file=f)
    if dump_wav:
        wdir = os.path.join(results_path, "pred_wav")
        os.makedirs(wdir, exist_ok=True)
        for n, i in enumerate(ids):
            w = hyps[i]["wav"]
            if w is not None:
                frontend.write_wav(os.path.join(wdir, f"{n}_pred.wav"), w.detach().cpu().numpy(), 16000)


def load_manifest(path: str, targets: Optional[Dict[int, List[int]]] = None) -> List[Tuple[int, str]]:
    """fairseq S2T/S2S manifest (TSV with header, columns `id` and `audio` = src_audio): one WAV per row.
    The sample id fairseq prints is the row index.  With `targets` given, the target units of the `tgt_audio` column
    (space-separated ids, SpeechToSpeechDataset: fairseq/data/audio/speech_to_speech_dataset.py) are collected per id."""
    rows = []
    with open(path, encoding="utf-8") as f:
        header = f.readline().rstrip("\n").split("\t")
        col = header.index("src_audio") if "src_audio" in header else header.index("audio")
        tcol = header.index("tgt_audio") if "tgt_audio" in header else -1
        for i, line in enumerate(f):
            parts = line.rstrip("\n").split("\t")
            if len(parts) > col and parts[col]:
                rows.append((i, parts[col]))
                if targets is not None and 0 <= tcol < len(parts):
                    try:
                        targets[i] = [int(u) for u in parts[tcol].split()]
                    except ValueError:      # a wav path, not units (S2ST with spectrogram targets)
                        pass
    return rows


def main(argv: Optional[List[str]] = None):
    from .agent import StreamSpeechS2STAgent
    from .modules import CodeHiFiGANVocoderWithDur, StreamSpeechModel, load_model_state
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("data", nargs="?", default=None, help="data root holding <gen-subset>.tsv and the config yamls")
    ap.add_argument("--gen-subset", default="test")
    ap.add_argument("--path", required=True, help="checkpoint (.pt) or synthetic:<seed>")
    ap.add_argument("--vocoder", required=True)
    ap.add_argument("--vocoder-cfg", default=None)
    ap.add_argument("--config-yaml", default=None)
    ap.add_argument("--multitask-config-yaml", default=None)
    ap.add_argument("--results-path", required=True)
    ap.add_argument("--wav-list", default=None, help="text file with one WAV path per line (instead of a manifest)")
    ap.add_argument("--synthetic", type=int, default=0, help="N synthetic utterances instead of files")
    ap.add_argument("--batch-size", type=int, default=32)
    ap.add_argument("--max-tokens", type=int, default=0)
    ap.add_argument("--max-len-a", type=float, default=0.0, help="unit generator (kept for CLI parity; the NAR decoder has no length search)")
    ap.add_argument("--max-len-b", type=int, default=200)
    ap.add_argument("--max-len-a-mt", type=float, default=0.0, help="first-pass text search: max_len = a * src_len + b")
    ap.add_argument("--max-len-b-mt", type=int, default=200)
    ap.add_argument("--dur-prediction", action="store_true")
    ap.add_argument("--no-wav", action="store_true")
    ap.add_argument("--scores", action="store_true")
    ap.add_argument("--num-shards", type=int, default=int(os.environ.get("WORLD_SIZE", "1")))
    ap.add_argument("--shard-id", type=int, default=int(os.environ.get("RANK", "0")))
    ap.add_argument("--device", default="cuda:%s" % os.environ.get("LOCAL_RANK", "0"))
    a = ap.parse_args(argv)

    # model / dictionaries / CMVN exactly as the agent loads them (agent :355-420)
    ns = argparse.Namespace(config_yaml=a.config_yaml, multitask_config_yaml=a.multitask_config_yaml,
                            data_bin=a.data or ".", model_path=a.path, global_stats=None, source_segment_size=999999 * 40,
                            shift_size=10, window_size=25, sample_rate=16000, feature_dim=80, full_recompute_encoder=True)
    holder = argparse.Namespace(device=a.device)
    StreamSpeechS2STAgent.load_model_vocab(holder, ns)
    model = holder.model.hip
    vcfg = None
    if a.vocoder_cfg:
        import json
        with open(a.vocoder_cfg) as f:
            vcfg = json.load(f)
    voc = CodeHiFiGANVocoderWithDur(a.vocoder, vcfg, device=a.device).hip

    targets: Dict[int, List[int]] = {}
    if a.synthetic > 0:
        from . import workload, synth
        utts = workload.make_utterances(a.synthetic)
        entries = [(u.idx, torch.from_numpy(synth.synth_pcm(1234 + u.idx, u.n_samples))) for u in utts]
    else:
        if a.wav_list:
            with open(a.wav_list) as f:
                rows = [(i, ln.strip()) for i, ln in enumerate(f) if ln.strip()]
        else:
            rows = load_manifest(os.path.join(a.data, a.gen_subset + ".tsv"), targets)
        entries = []
        for i, path in rows:
            x, sr = frontend.read_wav(path)
            entries.append((i, torch.from_numpy(x), sr))
    entries = entries[a.shard_id::a.num_shards]
    items = []
    for e in entries:
        pcm = e[1].to(a.device)
        if len(e) > 2 and e[2] != 16000:
            pcm = model.resample(pcm, e[2], 16000)
        items.append((e[0], pcm))
    sub = a.gen_subset if a.num_shards == 1 else f"{a.gen_subset}.shard{a.shard_id}"
    hyps = generate(model, voc, items, holder.dict, a.results_path, sub, a.batch_size, a.max_tokens, a.max_len_a,
                    a.max_len_b, a.max_len_a_mt, a.max_len_b_mt, a.dur_prediction, not a.no_wav,
                    getattr(holder.model, "uni_encoder", False), a.scores, targets=targets or None)
    print(f"| generated {len(hyps)} utterances into {a.results_path}", file=sys.stderr)


if __name__ == "__main__":
    main()
