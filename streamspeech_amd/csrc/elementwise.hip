// Row-wise / pointwise kernels (see elementwise.hpp for the reference citations).
#include "elementwise.hpp"

namespace ss {

// ---------------------------------------------------------------------------------------------
// LayerNorm: one wave64 per row, values kept in registers (D/64 per lane), two-pass statistics.
// ---------------------------------------------------------------------------------------------
template <int PER_LANE>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, int ldx, float* y,
                                                        int ldy, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, int M, float eps) {
  constexpr int D = PER_LANE * 64;
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const float* xr = x + (size_t)row * ldx;
  float v[PER_LANE];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < PER_LANE; ++i) { v[i] = xr[lane + 64 * i]; s += v[i]; }
  const float mean = wave_sum(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < PER_LANE; ++i) { const float d = v[i] - mean; q += d * d; }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)D + eps);
  float* yr = y + (size_t)row * ldy;
#pragma unroll
  for (int i = 0; i < PER_LANE; ++i) {
    const int c = lane + 64 * i;
    yr[c] = (v[i] - mean) * rstd * gamma[c] + beta[c];
  }
}

int launch_layernorm(const float* x, int ldx, float* y, int ldy, const float* gamma, const float* beta,
                     int M, int D, float eps, hipStream_t stream) {
  if (M <= 0) return SS_OK;
  dim3 grid(cdiv(M, 4)), block(256);
  switch (D) {
    case 64: hipLaunchKernelGGL(layernorm_kernel<1>, grid, block, 0, stream, x, ldx, y, ldy, gamma, beta, M, eps); break;
    case 128: hipLaunchKernelGGL(layernorm_kernel<2>, grid, block, 0, stream, x, ldx, y, ldy, gamma, beta, M, eps); break;
    case 256: hipLaunchKernelGGL(layernorm_kernel<4>, grid, block, 0, stream, x, ldx, y, ldy, gamma, beta, M, eps); break;
    case 512: hipLaunchKernelGGL(layernorm_kernel<8>, grid, block, 0, stream, x, ldx, y, ldy, gamma, beta, M, eps); break;
    case 1024: hipLaunchKernelGGL(layernorm_kernel<16>, grid, block, 0, stream, x, ldx, y, ldy, gamma, beta, M, eps); break;
    default: return SS_ERR_ARG;
  }
  SS_LAUNCH_CHECK();
  return SS_OK;
}

// ---------------------------------------------------------------------------------------------
// Chunk-causal depthwise conv (k = 31) + BatchNorm(eval) + SiLU, in LDS
// (reference chunk_unity/modules/conformer_layer.py:94-119 + chunk_causal_conv1d.py:39-68).
// Workgroup = 32 output rows x 64 channels: the (32 + K - 1)-row input slab is read from global once
// (coalesced 256-B row segments) into LDS, then thread (c, row group) slides over its 8 rows with
// the 31 taps of its channel in registers -- every input element is fetched from HBM/L2 once per
// workgroup instead of once per tap.  Visibility: taps at input positions >= the output row's
// chunk end (or the sequence end) are zero, the closed form of the reference's unfold/pad/re-stitch.
// ---------------------------------------------------------------------------------------------
constexpr int DW_TT = 32, DW_TC = 64, DW_KMAX = 31;

__global__ __launch_bounds__(256) void dwconv_bn_silu_kernel(
    const float* __restrict__ x, int ldx, float* y, int ldy, const float* __restrict__ wt, int K,
    const float* __restrict__ bn_mean, const float* __restrict__ bn_var, const float* __restrict__ bn_gamma,
    const float* __restrict__ bn_beta, float bn_eps, int T, int C, int chunk, const int* __restrict__ segs,
    int t_begin) {
  __shared__ float slab[(DW_TT + DW_KMAX - 1) * DW_TC];
  if (segs) {   // ragged batch: {row_start, len} per utterance
    const int st = segs[2 * blockIdx.z];
    T = segs[2 * blockIdx.z + 1];
    x += (size_t)st * ldx; y += (size_t)st * ldy;
  }
  const int t0 = t_begin + blockIdx.y * DW_TT;       // first output row of the tile
  if (t0 >= T) return;
  const int c0 = blockIdx.x * DW_TC;
  const int tid = threadIdx.x, cl = tid & (DW_TC - 1), rg = tid >> 6;
  const int half = K / 2;
  const int rows = DW_TT + K - 1;                    // slab row s <-> input row t0 - half + s
  {
    // All 16 loads of a thread are issued before the first LDS store: with a run-time trip count the loop was one
    // load -> wait -> store per iteration, ~1 us each when the launch is a handful of workgroups (a single utterance, the tail
    // rows of a streaming call): 13-18 us per launch whatever the row count (profiles/r03_streaming_kernel_stats.csv).
    constexpr int NL = ((DW_TT + DW_KMAX - 1) * DW_TC + 255) / 256;
    float v[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int idx = tid + i * 256;
      const int sr = idx / DW_TC, cc = idx - sr * DW_TC;
      const int tin = t0 - half + sr;
      v[i] = (sr < rows && tin >= 0 && tin < T && c0 + cc < C) ? x[(size_t)tin * ldx + c0 + cc] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int idx = tid + i * 256;
      if (idx < (DW_TT + DW_KMAX - 1) * DW_TC) slab[idx] = v[i];
    }
  }
  const int c = c0 + cl;
  float w[DW_KMAX];
#pragma unroll
  for (int j = 0; j < DW_KMAX; ++j) w[j] = (j < K && c < C) ? wt[j * C + c] : 0.f;
  __syncthreads();
  if (c >= C) return;
  const float mean = bn_mean[c], rstd = 1.0f / sqrtf(bn_var[c] + bn_eps), gam = bn_gamma[c], bet = bn_beta[c];
  // This thread's 8 consecutive rows share their taps' inputs: slab rows rg*8 .. rg*8 + 37 of channel cl are read ONCE into
  // registers (38 LDS reads in flight together), then each row slides over them.  Branch-free: the visibility test selects
  // the result of every tap -- with `if (j < jmax)` around the fma each tap was its own read -> wait -> fma, 13 us per
  // launch whatever the row count (tools/dwconv_probe.py; 12 such launches per encoder call of a single utterance).
  constexpr int NR = DW_TT / 4, NV = NR + DW_KMAX - 1;
  float sv[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) sv[i] = slab[(rg * NR + i) * DW_TC + cl];
#pragma unroll
  for (int u = 0; u < NR; ++u) {
    const int t = t0 + rg * NR + u;
    int lim = T;                                     // first invisible input row
    if (chunk > 0) { const int cl_end = (t / chunk + 1) * chunk; if (cl_end < lim) lim = cl_end; }
    const int jmax = min(K, lim - (t - half));       // taps j < jmax are visible (input row t - half + j < lim)
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < DW_KMAX; ++j) {
      const float a2 = fmaf(w[j], sv[u + j], acc);
      acc = (j < jmax) ? a2 : acc;
    }
    const float v = (acc - mean) * rstd * gam + bet;
    if (t < T) y[(size_t)t * ldy + c] = v / (1.0f + expf(-v));
  }
}

int launch_dwconv_bn_silu(const float* x, int ldx, float* y, int ldy, const float* wt, int K,
                          const float* bn_mean, const float* bn_var, const float* bn_gamma,
                          const float* bn_beta, float bn_eps, int T, int C, int chunk, hipStream_t stream,
                          const int* segs, int nseg, int t_begin) {
  if (T - t_begin <= 0) return SS_OK;
  if (K > DW_KMAX || (K & 1) == 0) return SS_ERR_ARG;
  dim3 grid(cdiv(C, DW_TC), cdiv(T - t_begin, DW_TT), nseg > 0 ? nseg : 1);
  hipLaunchKernelGGL(dwconv_bn_silu_kernel, grid, dim3(256), 0, stream, x, ldx, y, ldy, wt, K, bn_mean,
                     bn_var, bn_gamma, bn_beta, bn_eps, T, C, chunk, nseg > 0 ? segs : nullptr, t_begin);
  SS_LAUNCH_CHECK();
  return SS_OK;
}

// ---------------------------------------------------------------------------------------------
// Embeddings
// ---------------------------------------------------------------------------------------------
__global__ void embed_tokens_kernel(const int* __restrict__ tok, const float* __restrict__ emb,
                                    const float* __restrict__ pos_table, float scale, int pos0, float* out,
                                    int n, int D, int pos_stride, int pad_id, int vocab) {
  const int i = blockIdx.y;
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= D) return;
  int tk = tok[i];
  if ((unsigned)tk >= (unsigned)vocab) tk = 0;            // ids come from device memory (the token chain): never index out of the table
  // make_positions (fairseq/utils.py:256-266): a <pad> token takes position padding_idx (the zero row)
  const int pos = (tk == pad_id) ? pad_id : pos0 + i * pos_stride;
  out[(size_t)i * D + c] = scale * emb[(size_t)tk * D + c] + pos_table[(size_t)pos * D + c];
}

int launch_embed_tokens(const int* tok, const float* emb, const float* pos_table, float scale, int pos0,
                        float* out, int n, int D, hipStream_t stream, int pos_stride, int pad_id, int vocab) {
  if (n <= 0) return SS_OK;
  hipLaunchKernelGGL(embed_tokens_kernel, dim3(cdiv(D, 256), n), dim3(256), 0, stream, tok, emb, pos_table,
                     scale, pos0, out, n, D, pos_stride, pad_id, vocab);
  SS_LAUNCH_CHECK();
  return SS_OK;
}

__global__ void upsample_add_pos_kernel(const float* __restrict__ src, int up, const float* __restrict__ pos_row,
                                        float pad_value, float* out, int D) {
  const int u = blockIdx.y;
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= D) return;
  const float* s = src + (size_t)(u / up) * D;
  const float add = (s[0] != pad_value) ? pos_row[c] : 0.f;
  out[(size_t)u * D + c] = s[c] + add;
}

int launch_upsample_add_pos(const float* src, int n, int up, const float* pos_row, float pad_value,
                            float* out, int D, hipStream_t stream) {
  if (n <= 0) return SS_OK;
  hipLaunchKernelGGL(upsample_add_pos_kernel, dim3(cdiv(D, 256), n * up), dim3(256), 0, stream, src, up,
                     pos_row, pad_value, out, D);
  SS_LAUNCH_CHECK();
  return SS_OK;
}

__global__ void gather_rows_kernel(const int* __restrict__ idx, const float* __restrict__ table, int D,
                                   float* out, int rows) {
  const int i = blockIdx.y;
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= D) return;
  int r = idx[i];
  if ((unsigned)r >= (unsigned)rows) r = 0;                // ids come from device memory: never read outside the table
  out[(size_t)i * D + c] = table[(size_t)r * D + c];
}

int launch_gather_rows(const int* idx, const float* table, int D, float* out, int n, hipStream_t stream, int rows) {
  if (n <= 0) return SS_OK;
  hipLaunchKernelGGL(gather_rows_kernel, dim3(cdiv(D, 256), n), dim3(256), 0, stream, idx, table, D, out, rows);
  SS_LAUNCH_CHECK();
  return SS_OK;
}

// ---------------------------------------------------------------------------------------------
// Masked argmax over a row (CTC heads, unit head, MT next-token).  log_softmax is monotone, so the
// reference's log_softmax -> set -inf -> max (agent/ctc_decoder.py:52-60) equals an argmax of the
// logits with those indices skipped; ties resolve to the lowest index like torch.max / topk.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void masked_argmax_kernel(const float* __restrict__ logits, int ld, int M, int N,
                                                            int mask0, int mask1, int mask2, int force, int* ids,
                                                            const int* __restrict__ row_max_len, int step, int force_id) {
  // one workgroup per row: 256 threads stride over the vocabulary, wave shuffle + LDS reduce
  __shared__ float sb[4];
  __shared__ int si[4];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int row = blockIdx.x;
  if (row_max_len && step >= row_max_len[row]) force = force_id;   // batched search: per-utterance max length
  if (force >= 0) { if (t == 0) ids[row] = force; return; }
  const float* r = logits + (size_t)row * ld;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int n = t; n < N; n += 256) {
    if (n == mask0 || n == mask1 || n == mask2) continue;
    float v = r[n];
    if (v != v) v = -INFINITY;  // NaN -> -inf (agent/sequence_generator.py:350); still a candidate, so a row of NaNs yields the
                                // first unmasked column like an all -inf row does there, never an out-of-range id
    if (bi == 0x7fffffff || v > best) { best = v; bi = n; }  // n ascends per thread: first max wins
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ob = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (oi != 0x7fffffff && (bi == 0x7fffffff || ob > best || (ob == best && oi < bi))) { best = ob; bi = oi; }
  }
  if (lane == 0) { sb[wave] = best; si[wave] = bi; }
  __syncthreads();
  if (t == 0) {
    for (int w = 1; w < 4; ++w) {
      const float ob = sb[w]; const int oi = si[w];
      if (oi != 0x7fffffff && (bi == 0x7fffffff || ob > best || (ob == best && oi < bi))) { best = ob; bi = oi; }
    }
    ids[row] = bi;
  }
}

int launch_masked_argmax(const float* logits, int ld, int M, int N, int mask0, int mask1, int mask2,
                         int force, int* ids, hipStream_t stream, const int* row_max_len, int step, int force_id) {
  if (M <= 0) return SS_OK;
  hipLaunchKernelGGL(masked_argmax_kernel, dim3(M), dim3(256), 0, stream, logits, ld, M, N, mask0,
                     mask1, mask2, force, ids, row_max_len, step, force_id);
  SS_LAUNCH_CHECK();
  return SS_OK;
}

// ---------------------------------------------------------------------------------------------
// Per-row maximum log-probability: out[m] = max_{n not masked} logits[m,n] - logsumexp_n logits[m,:]
// = what `lprobs = log_softmax(logits); lprobs[:, masked] = -inf; lprobs.max(-1)` gives
// (researches/ctc_unity/ctc_generator.py:55-63: the score / positional_scores of fairseq-generate's H- / P- lines).
// One workgroup per row; two passes (max, then sum of exp) in f32 like torch's log_softmax.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void row_max_logprob_kernel(const float* __restrict__ logits, int ld, int N,
                                                              int mask0, int mask1, int mask2, float* __restrict__ out) {
  __shared__ float sa[4], sb[4];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const float* r = logits + (size_t)blockIdx.x * ld;
  float mx = -INFINITY, best = -INFINITY;
  for (int n = t; n < N; n += 256) {
    const float v = r[n];
    mx = fmaxf(mx, v);
    if (n != mask0 && n != mask1 && n != mask2) best = fmaxf(best, v);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { mx = fmaxf(mx, __shfl_xor(mx, o, 64)); best = fmaxf(best, __shfl_xor(best, o, 64)); }
  if (lane == 0) { sa[wave] = mx; sb[wave] = best; }
  __syncthreads();
  mx = fmaxf(fmaxf(sa[0], sa[1]), fmaxf(sa[2], sa[3]));
  best = fmaxf(fmaxf(sb[0], sb[1]), fmaxf(sb[2], sb[3]));
  __syncthreads();
  float sum = 0.f;
  for (int n = t; n < N; n += 256) sum += expf(r[n] - mx);     // accurate forms: off the timed path, printed to 4+ decimals
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
  if (lane == 0) sa[wave] = sum;
  __syncthreads();
  if (t == 0) out[blockIdx.x] = (best - mx) - logf(sa[0] + sa[1] + sa[2] + sa[3]);
}

// Full log-softmax / softmax rows (lprobs the agents hand back with --output-asr-translation style consumers, and
// model.get_normalized_probs): out[r, n] = x - max - log(sum exp(x - max)), ids mask0 / mask1 set to -inf AFTER the
// normalisation (agent/ctc_decoder.py:52-60).  Workgroup per row, accurate expf / logf: off the timed path.
__global__ __launch_bounds__(256) void log_softmax_kernel(const float* __restrict__ logits, int ld, int N, int mask0, int mask1,
                                                          int as_probs, float* __restrict__ out, int ldo) {
  __shared__ float sa[4];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const float* r = logits + (size_t)blockIdx.x * ld;
  float* o = out + (size_t)blockIdx.x * ldo;
  float mx = -INFINITY;
  for (int n = t; n < N; n += 256) mx = fmaxf(mx, r[n]);
  mx = wave_max(mx);
  if (lane == 0) sa[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(sa[0], sa[1]), fmaxf(sa[2], sa[3]));
  __syncthreads();
  float sum = 0.f;
  for (int n = t; n < N; n += 256) sum += expf(r[n] - mx);
  sum = wave_sum(sum);
  if (lane == 0) sa[wave] = sum;
  __syncthreads();
  const float tot = (sa[0] + sa[1]) + (sa[2] + sa[3]);
  const float lse = logf(tot);
  for (int n = t; n < N; n += 256) {
    float v = as_probs ? expf(r[n] - mx) / tot : (r[n] - mx) - lse;
    if (n == mask0 || n == mask1) v = as_probs ? 0.f : -INFINITY;
    o[n] = v;
  }
}

int launch_log_softmax(const float* logits, int ld, int M, int N, int mask0, int mask1, int as_probs, float* out, int ldo,
                       hipStream_t stream) {
  if (M <= 0) return SS_OK;
  hipLaunchKernelGGL(log_softmax_kernel, dim3(M), dim3(256), 0, stream, logits, ld, N, mask0, mask1, as_probs, out, ldo);
  SS_LAUNCH_CHECK();
  return SS_OK;
}

int launch_row_max_logprob(const float* logits, int ld, int M, int N, int mask0, int mask1, int mask2, float* out,
                           hipStream_t stream) {
  if (M <= 0) return SS_OK;
  hipLaunchKernelGGL(row_max_logprob_kernel, dim3(M), dim3(256), 0, stream, logits, ld, N, mask0, mask1, mask2, out);
  SS_LAUNCH_CHECK();
  return SS_OK;
}

// ---------------------------------------------------------------------------------------------
// CTC collapse: single workgroup, chunks of 1024 frames with a running output offset.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void ctc_collapse_kernel(const int* __restrict__ raw, int T, int blank, int pad,
                                                            int* tokens, int* index, int* count,
                                                            const int* __restrict__ segs) {
  __shared__ int wave_tot[16];
  __shared__ int base_s;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  if (segs) {   // one workgroup per utterance: rows [start, start+len) of the packed arrays
    const int st = segs[2 * blockIdx.x];
    T = segs[2 * blockIdx.x + 1];
    raw += st; tokens += st; index += st; count += blockIdx.x;
  }
  if (t == 0) base_s = 0;
  __syncthreads();
  for (int c0 = 0; c0 < T; c0 += 1024) {
    const int i = c0 + t;
    int v = 0;
    bool keep = false;
    if (i < T) {
      v = raw[i];
      keep = (i == 0 || v != raw[i - 1]) && v != blank && v != pad;
    }
    const unsigned long long bal = __ballot(keep);
    const int pre = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) wave_tot[wave] = __popcll(bal);
    __syncthreads();
    int off = base_s;
    for (int w = 0; w < wave; ++w) off += wave_tot[w];
    if (keep) { tokens[off + pre] = v; index[off + pre] = i; }
    __syncthreads();
    if (t == 0) { int s = 0; for (int w = 0; w < 16; ++w) s += wave_tot[w]; base_s += s; }
    __syncthreads();
  }
  if (t == 0) *count = base_s;
}

int launch_ctc_collapse(const int* raw, int T, int blank, int pad, int* tokens, int* index, int* count,
                        hipStream_t stream, const int* segs, int nseg) {
  hipLaunchKernelGGL(ctc_collapse_kernel, dim3(nseg > 0 ? nseg : 1), dim3(1024), 0, stream, raw, T, blank, pad, tokens,
                     index, count, nseg > 0 ? segs : nullptr);
  SS_LAUNCH_CHECK();
  return SS_OK;
}

// ---------------------------------------------------------------------------------------------
// Duration predictor tail + repeat_interleave
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void dur_predict_kernel(const float* __restrict__ logdur,
                                                           const int* __restrict__ forced, int K, int* dur, int* cum,
                                                           const int* __restrict__ segs) {
  __shared__ int wave_tot[16];
  __shared__ int base_s;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  if (segs) {   // utterance s: units [start, start+len); its cum[] (len+1 entries) starts at start + s
    const int st = segs[2 * blockIdx.x];
    K = segs[2 * blockIdx.x + 1];
    logdur += st; dur += st; cum += st + blockIdx.x;
    if (forced) forced += st;
  }
  if (t == 0) { base_s = 0; cum[0] = 0; }
  __syncthreads();
  for (int c0 = 0; c0 < K; c0 += 1024) {
    const int k = c0 + t;
    int d = 0;
    if (k < K) {
      if (forced) d = forced[k];
      else {
        // torch.round is round-half-to-even == rintf in the default rounding mode
        const float r = rintf(expf(logdur[k]) - 1.0f);
        d = (int)fmaxf(r, 1.0f);
      }
      dur[k] = d;
    }
    int incl = d;  // inclusive scan inside the wave
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int n = __shfl_up(incl, o, 64); if (lane >= o) incl += n; }
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    int off = base_s;
    for (int w = 0; w < wave; ++w) off += wave_tot[w];
    if (k < K) cum[k + 1] = off + incl;
    __syncthreads();
    if (t == 0) { int s = 0; for (int w = 0; w < 16; ++w) s += wave_tot[w]; base_s += s; }
    __syncthreads();
  }
}

int launch_dur_predict(const float* logdur, const int* forced, int K, int* dur, int* cum, hipStream_t stream,
                       const int* segs, int nseg) {
  if (K <= 0 && nseg <= 0) return SS_ERR_ARG;
  hipLaunchKernelGGL(dur_predict_kernel, dim3(nseg > 0 ? nseg : 1), dim3(1024), 0, stream, logdur, forced, K, dur, cum,
                     nseg > 0 ? segs : nullptr);
  SS_LAUNCH_CHECK();
  return SS_OK;
}

__global__ void repeat_rows_kernel(const float* __restrict__ emb, const int* __restrict__ cum, int K, int D,
                                   float* out, const int* __restrict__ segs) {
  const int f = blockIdx.y;
  if (segs) {   // {unit_start, n_units, frame_start, n_frames}
    const int* sg = segs + 4 * blockIdx.z;
    if (f >= sg[3]) return;
    emb += (size_t)sg[0] * D; cum += sg[0] + blockIdx.z; K = sg[1]; out += (size_t)sg[2] * D;
  }
  // largest k with cum[k] <= f
  int lo = 0, hi = K - 1;
  while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (cum[mid] <= f) lo = mid; else hi = mid - 1; }
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c < D) out[(size_t)f * D + c] = emb[(size_t)lo * D + c];
}

int launch_repeat_rows(const float* emb, const int* cum, int K, int D, float* out, int F, hipStream_t stream,
                       const int* segs, int nseg) {
  if (F <= 0) return SS_OK;
  hipLaunchKernelGGL(repeat_rows_kernel, dim3(cdiv(D, 256), F, nseg > 0 ? nseg : 1), dim3(256), 0, stream, emb, cum, K, D,
                     out, nseg > 0 ? segs : nullptr);
  SS_LAUNCH_CHECK();
  return SS_OK;
}

// ---------------------------------------------------------------------------------------------
// HiFi-GAN conv_post (C -> 1, k = 7) + tanh.  HBM/L2-bound: thread = output sample, reads a
// contiguous 7*C window of the channels-last input.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void conv_post_tanh_kernel(const float* __restrict__ x, int T, int C,
                                                             const float* __restrict__ w, const float* __restrict__ bias,
                                                             float slope, float* wav, const int* __restrict__ segs) {
  extern __shared__ float ws[];  // 7*C weights
  for (int i = threadIdx.x; i < 7 * C; i += 256) ws[i] = w[i];
  __syncthreads();
  if (segs) {   // {sample_start, n_samples}
    const int st = segs[2 * blockIdx.y];
    T = segs[2 * blockIdx.y + 1];
    x += (size_t)st * C; wav += st;
  }
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= T) return;
  float acc = 0.f;
  for (int j = 0; j < 7; ++j) {
    const int pos = t + j - 3;
    if (pos < 0 || pos >= T) continue;
    const float* xr = x + (size_t)pos * C;
    for (int c = 0; c < C; ++c) {
      float v = xr[c];
      v = v > 0.f ? v : v * slope;
      acc = fmaf(ws[j * C + c], v, acc);
    }
  }
  wav[t] = tanhf(acc + bias[0]);
}

int launch_conv_post_tanh(const float* x, int T, int C, const float* w, const float* bias, float slope,
                          float* wav, hipStream_t stream, const int* segs, int nseg) {
  if (T <= 0) return SS_OK;
  hipLaunchKernelGGL(conv_post_tanh_kernel, dim3(cdiv(T, 256), nseg > 0 ? nseg : 1), dim3(256), 7 * C * sizeof(float),
                     stream, x, T, C, w, bias, slope, wav, nseg > 0 ? segs : nullptr);
  SS_LAUNCH_CHECK();
  return SS_OK;
}

}  // namespace ss
