// Fused fbank + CMVN front-end kernel (see fbank.hip).
#pragma once
#include "common.hpp"

namespace ss {

// pcm: mono 16 kHz float samples on the device; each sample is multiplied by pcm_scale (2^15 for
// [-1,1] input, reference fairseq/examples/speech_to_text/data_utils.py:85).  window [400] and
// melw [80][257] are host-built constants living in the weight blob.  Writes feat [T,80] and
// returns T = 1 + (n-400)/160 (snip_edges) through n_frames (host int).
int launch_fbank_cmvn(const float* pcm, int n_samples, float pcm_scale, const float* window,
                      const float* melw, const float* cmvn_mean, const float* cmvn_std, float* feat,
                      int* n_frames, hipStream_t stream);

// Ragged batch: segs[3*s] = {pcm_start, n_frames, frame_start} into the packed PCM / feature arrays.
int launch_fbank_cmvn_batch(const float* pcm, float pcm_scale, const float* window, const float* melw,
                            const float* cmvn_mean, const float* cmvn_std, float* feat, const int* segs, int nseg,
                            int max_frames, hipStream_t stream);

// y[k] = sum_m x[m] * taps[half_len + k*down - m*up] for k < n_out (zero-phase polyphase FIR; `up`/`down`
// in lowest terms, taps [2*half_len+1] on the device with gain `up`).
int launch_resample(const float* x, long long n_in, int up, int down, const float* taps, int half_len, float* y,
                    long long n_out, hipStream_t stream);

}  // namespace ss
