// Fused Kaldi-compatible fbank + global CMVN.
//
// Replaces OnlineFeatureExtractor -> torchaudio.compliance.kaldi.fbank -> (x-mean)/std
// (reference agent/speech_to_speech.streamspeech.agent.py:66-98, fairseq/data/audio/audio_utils.py:236-249)
// with ONE kernel: workgroup = one 25 ms frame, everything between the coalesced PCM read and the
// 80-float feature row write stays in LDS: DC removal, pre-emphasis 0.97, Povey window, 512-point
// radix-2 FFT, power spectrum, 80 triangular mel bins, log floor, CMVN.
#include "fbank.hpp"

namespace ss {

constexpr int WIN = 400, SHIFT = 160, NFFT = 512, NBIN = 257, NMEL = 80;

__global__ __launch_bounds__(256) void fbank_cmvn_kernel(const float* __restrict__ pcm, float pcm_scale,
                                                         const float* __restrict__ window,   // [400]
                                                         const float* __restrict__ melw,     // [80][257]
                                                         const float* __restrict__ cmvn_mean,
                                                         const float* __restrict__ cmvn_std, float* feat,
                                                         const int* __restrict__ segs) {
  __shared__ float re[NFFT], im[NFFT];
  __shared__ float tw_c[NFFT / 2], tw_s[NFFT / 2];
  __shared__ float red[4];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int frame = blockIdx.x;
  if (segs) {   // ragged batch {pcm_start, n_frames, frame_start}; whole workgroup exits together
    const int* sg = segs + 3 * blockIdx.y;
    if (frame >= sg[1]) return;
    pcm += sg[0]; feat += (size_t)sg[2] * NMEL;
  }
  const float* src = pcm + (size_t)frame * SHIFT;

  // load (coalesced) and frame mean
  const float x0 = src[t] * pcm_scale;
  const float x1 = (t + 256 < WIN) ? src[t + 256] * pcm_scale : 0.f;
  float s = wave_sum(x0 + x1);
  if (lane == 0) red[wave] = s;
  // twiddles e^{-2 pi i k / 512}
  {
    float sn, cs;
    sincospif(-2.0f * (float)t / (float)NFFT, &sn, &cs);
    tw_c[t] = cs; tw_s[t] = sn;
  }
  __syncthreads();
  const float mean = (red[0] + red[1] + red[2] + red[3]) / (float)WIN;
  re[t] = x0 - mean;
  re[t + 256] = (t + 256 < WIN) ? (x1 - mean) : 0.f;
  __syncthreads();
  // pre-emphasis (replicate-pad on the left) + window, written in bit-reversed order for the DIT FFT
  float y0, y1 = 0.f;
  {
    const float prev0 = re[t > 0 ? t - 1 : 0];
    y0 = (re[t] - 0.97f * prev0) * window[t];
    if (t + 256 < WIN) y1 = (re[t + 256] - 0.97f * re[t + 255]) * window[t + 256];
  }
  __syncthreads();
  re[__brev((unsigned)t) >> 23] = y0;
  re[__brev((unsigned)(t + 256)) >> 23] = y1;
  im[t] = 0.f; im[t + 256] = 0.f;
  __syncthreads();
  // 9 radix-2 stages, one butterfly per thread
#pragma unroll
  for (int stg = 0; stg < 9; ++stg) {
    const int half = 1 << stg;
    const int grp = t >> stg, pos = t & (half - 1);
    const int i0 = (grp << (stg + 1)) + pos, i1 = i0 + half;
    const int tk = pos << (8 - stg);
    const float c = tw_c[tk], sn = tw_s[tk];
    const float br = re[i1] * c - im[i1] * sn;
    const float bi = re[i1] * sn + im[i1] * c;
    const float ar = re[i0], ai = im[i0];
    __syncthreads();
    re[i0] = ar + br; im[i0] = ai + bi;
    re[i1] = ar - br; im[i1] = ai - bi;
    __syncthreads();
  }
  // power spectrum into re[0..256]
  const float p0 = re[t] * re[t] + im[t] * im[t];
  const float p256 = (t == 0) ? (re[256] * re[256] + im[256] * im[256]) : 0.f;
  __syncthreads();
  re[t] = p0;
  if (t == 0) re[256] = p256;
  __syncthreads();
  if (t < NMEL) {
    const float* w = melw + t * NBIN;
    float e = 0.f;
    for (int i = 0; i < NBIN; ++i) e = fmaf(w[i], re[i], e);
    const float lg = logf(fmaxf(e, 1.1920928955078125e-07f));
    feat[(size_t)frame * NMEL + t] = (lg - cmvn_mean[t]) / cmvn_std[t];
  }
}

int launch_fbank_cmvn(const float* pcm, int n_samples, float pcm_scale, const float* window,
                      const float* melw, const float* cmvn_mean, const float* cmvn_std, float* feat,
                      int* n_frames, hipStream_t stream) {
  const int T = n_samples < WIN ? 0 : 1 + (n_samples - WIN) / SHIFT;
  if (n_frames) *n_frames = T;
  if (T == 0) return SS_OK;
  hipLaunchKernelGGL(fbank_cmvn_kernel, dim3(T), dim3(256), 0, stream, pcm, pcm_scale, window, melw,
                     cmvn_mean, cmvn_std, feat, (const int*)nullptr);
  SS_LAUNCH_CHECK();
  return SS_OK;
}

int launch_fbank_cmvn_batch(const float* pcm, float pcm_scale, const float* window, const float* melw,
                            const float* cmvn_mean, const float* cmvn_std, float* feat, const int* segs, int nseg,
                            int max_frames, hipStream_t stream) {
  if (nseg <= 0 || max_frames <= 0) return SS_OK;
  hipLaunchKernelGGL(fbank_cmvn_kernel, dim3(max_frames, nseg), dim3(256), 0, stream, pcm, pcm_scale, window, melw,
                     cmvn_mean, cmvn_std, feat, segs);
  SS_LAUNCH_CHECK();
  return SS_OK;
}

// ---------------------------------------------------------------------------------------------
// Rational polyphase resampler (48 kHz -> 16 kHz in the agent).  Replaces the sox `rate` effect
// the reference applies to the whole sample history at every chunk
// (fairseq/data/audio/audio_utils.py:53-62 <- convert_waveform <- agent :66-98) with a zero-phase
// windowed-sinc FIR: y[k] = sum_m x[m] * h[half + k*down - m*up].  The taps (host-designed, gain
// `up`) are ~60/output sample for 3:1: a thread per output sample, taps through LDS when they fit.
// HBM-bound by construction: 4*(n_in + n_out) bytes.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void resample_kernel(const float* __restrict__ x, long long n_in, int up, int down,
                                                       const float* __restrict__ h, int half, float* __restrict__ y,
                                                       long long n_out) {
  extern __shared__ float hs[];
  const int ntaps = 2 * half + 1;
  for (int i = threadIdx.x; i < ntaps; i += 256) hs[i] = h[i];
  __syncthreads();
  const long long k = (long long)blockIdx.x * 256 + threadIdx.x;
  if (k >= n_out) return;
  const long long c = k * down;
  long long m_lo = c - half;                          // ceil((c - half) / up), clamped at 0
  m_lo = m_lo <= 0 ? 0 : (m_lo + up - 1) / up;
  long long m_hi = (c + half) / up;
  if (m_hi > n_in - 1) m_hi = n_in - 1;
  float acc = 0.f;
  for (long long m = m_lo; m <= m_hi; ++m) acc = fmaf(x[m], hs[half + (int)(c - m * up)], acc);
  y[k] = acc;
}

int launch_resample(const float* x, long long n_in, int up, int down, const float* taps, int half_len, float* y,
                    long long n_out, hipStream_t stream) {
  if (n_in <= 0 || n_out <= 0) return SS_OK;
  if (up <= 0 || down <= 0 || half_len < 0) return SS_ERR_ARG;
  const size_t lds = (size_t)(2 * half_len + 1) * sizeof(float);
  if (lds > 64 * 1024) return SS_ERR_ARG;             // 16 K taps: ratios up to ~800:1 in lowest terms
  hipLaunchKernelGGL(resample_kernel, dim3((unsigned)((n_out + 255) / 256)), dim3(256), lds, stream, x, n_in, up, down,
                     taps, half_len, y, n_out);
  SS_LAUNCH_CHECK();
  return SS_OK;
}

}  // namespace ss
