"""The offline driver on the HIP path against the reference's offline generator classes (fixture written by
oracle/make_golden_offline.py): A-/S-/D- text lines exact, T-/H-/D-/P- lines of generate-<subset>.txt with identical
unit strings, scores within f32 noise (the score is a sum over 25 (N+1) positions of max log-probabilities from
ss_row_max_logprob), wav files dumped under generate_waveform_from_code.py's names."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_offline_driver_hip_equals_reference_generator_lines(hip_model, hip_vocoder, tmp_path):
    from tests import offline_fixture as OF
    r = OF.run_and_compare(hip_model, hip_vocoder, hip_model.cfg, "short_search", tmp_path, device="cuda:0",
                           score_rel=2e-5, pos_abs=1e-3)
    print("offline generator parity (HIP vs reference classes):", {k: r[k] for k in ("worst_score_rel", "worst_pos_abs")})
    assert (tmp_path / "pred_wav" / "0_pred.wav").exists()
    # one utterance per batch gives the same files as the ragged batch of four
    r1 = OF.run_and_compare(hip_model, None, hip_model.cfg, "short_search", tmp_path / "b1", device="cuda:0", batch_size=1,
                            score_rel=2e-5, pos_abs=1e-3)
    assert all(r1["hyps"][i]["units"] == r["hyps"][i]["units"] for i in r["ids"])


def test_offline_driver_hip_default_search_length(hip_model, tmp_path):
    from tests import offline_fixture as OF
    OF.run_and_compare(hip_model, None, hip_model.cfg, "default_search", tmp_path, device="cuda:0", score_rel=2e-5, pos_abs=1e-3)


def test_row_max_logprob_kernel_matches_torch(hip_model):
    import ctypes as C
    import torch
    from streamspeech_amd import lib as L
    lib = L.load()
    torch.manual_seed(3)
    for rows, V in ((1, 1005), (77, 1005), (5, 6000), (3, 64)):
        x = (torch.randn(rows, V) * 6).cuda()
        out = torch.empty(rows, device="cuda")
        L.check(lib.ss_row_max_logprob(C.c_void_p(torch.cuda.current_stream().cuda_stream), C.c_void_p(x.data_ptr()), rows, V, 1, 3, 2,
                                       C.c_void_p(out.data_ptr())), "ss_row_max_logprob")
        lp = torch.log_softmax(x.cpu().double(), -1)
        lp[:, [1, 2, 3]] = -np.inf
        assert (out.cpu().double() - lp.max(-1).values).abs().max() < 2e-5
