"""Activation tensors past 2^31 BYTES in the narrow vocoder stages: a pack whose 16-channel stage is 2.2 GB per
tensor stays on the slab kernels (64-bit row addressing) and gives the waveforms the same utterances get in two smaller packs.
Rounds 1-5 sent such tensors to the generic LDS tiles on a byte bound inherited from the buffer-addressed stream-K kernels, which
capped the pack size at 128 utterances.  Reference op: fairseq/models/text_to_speech/hifigan.py:95-102, 154-170."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu


def _launches(lib, name):
    for c in range(lib.ss_prof_num_classes()):
        if lib.ss_prof_class_name(c).decode() == name:
            n = C.c_int64()
            lib.ss_prof_totals(c, None, None, C.byref(n))
            return n.value
    raise KeyError(name)


def test_vocoder_pack_with_2_2_gb_stage_tensors_stays_on_the_slab_kernels():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    if torch.cuda.mem_get_info()[0] < (60 << 30):
        pytest.skip("needs ~40 GB of free HBM")
    from streamspeech_amd import synth
    from streamspeech_amd.config import VocoderConfig
    from streamspeech_amd.engine import HipVocoder, Scratch
    vcfg = VocoderConfig()
    voc = HipVocoder(synth.make_vocoder_state_dict(0, vcfg), vcfg, scratch=Scratch())
    lib = voc.lib
    B, K = 36, 1000
    codes = [[int(x) for x in synth.uniform(900 + b, "big_pack_units", (K,), 0, 1000)] for b in range(B)]
    durs = [[3] * K for _ in range(B)]                       # 36 x 3000 frames = 108 000 frames -> 34.56 M rows of 16 channels = 2.21 GB
    rows16 = B * 3 * K * 320
    assert rows16 * 16 * 4 > 2 ** 31
    halves = []
    for part in (slice(0, B // 2), slice(B // 2, B)):
        w, _, _ = voc.batch_forward(codes[part], True, durs[part])
        halves += [x.clone() for x in w]
    torch.cuda.synchronize()
    tiles0 = _launches(lib, "conv_gemm<128,16,16,4,1>") + _launches(lib, "conv_gemm<128,32,32,4,1>") + _launches(lib, "conv_gemm<128,32,16,4,1>")
    slab0 = _launches(lib, "conv_c16<256,16>") + _launches(lib, "resblock_fused<16>")
    wavs, _, _ = voc.batch_forward(codes, True, durs)
    torch.cuda.synchronize()
    assert _launches(lib, "conv_gemm<128,16,16,4,1>") + _launches(lib, "conv_gemm<128,32,32,4,1>") + _launches(lib, "conv_gemm<128,32,16,4,1>") == tiles0, \
        "a narrow-stage conv of the big pack went to the generic tiles"
    assert _launches(lib, "conv_c16<256,16>") + _launches(lib, "resblock_fused<16>") > slab0
    worst = 0.0
    for a, b in zip(wavs, halves):
        assert a.numel() == b.numel() == 3 * K * 320 and torch.isfinite(a).all()
        worst = max(worst, float(torch.sqrt(torch.mean((a - b) ** 2))))
    print(f"2.2-GB stage tensors: worst waveform RMS between the 36-utterance pack and its two halves {worst:.2e}")
    assert worst < 2e-6
    voc.scratch.trim(0)
