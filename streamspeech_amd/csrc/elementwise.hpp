// Row-wise / pointwise kernels of the S2ST path (LayerNorm, chunk-causal depthwise conv + BN + SiLU,
// embeddings, masked argmax, CTC collapse, vocoder glue).
#pragma once
#include "common.hpp"

namespace ss {

// y[m,:] = LayerNorm(x[m,:]) * gamma + beta   (eps 1e-5; torch.nn.LayerNorm semantics:
// biased variance of deviations from the mean).  D multiple of 64, D <= 1024.
int launch_layernorm(const float* x, int ldx, float* y, int ldy, const float* gamma, const float* beta,
                     int M, int D, float eps, hipStream_t stream);

// Conformer ConvolutionModule middle (reference chunk_unity/modules/conformer_layer.py:108-113):
// y[t,c] = SiLU(BN_eval(sum_j w[j,c] * x[t+j-K/2, c]))  with the ChunkCausalConv1d visibility rule
// (positions >= (t/chunk+1)*chunk and outside [0,T) read as zero; chunk = 0 -> plain "same" conv).
// wt is the depthwise weight transposed to [K][C].
int launch_dwconv_bn_silu(const float* x, int ldx, float* y, int ldy, const float* wt, int K,
                          const float* bn_mean, const float* bn_var, const float* bn_gamma,
                          const float* bn_beta, float bn_eps, int T, int C, int chunk, hipStream_t stream,
                          const int* segs = nullptr, int nseg = 0,   // segs {row_start,len}; T = max len
                          int t_begin = 0);   // single utterance: only rows t_begin..T-1 are computed (rows before are context)

// out[i,:] = scale * emb[tok[i],:] + pos_table[pos0 + i, :]     (MT decoder input embedding,
// reference ctc_unity/modules/transformer_decoder.py:297-326)
int launch_embed_tokens(const int* tok, const float* emb, const float* pos_table, float scale, int pos0,
                        float* out, int n, int D, hipStream_t stream, int pos_stride, int pad_id, int vocab);
// pos_stride 0: same position for all rows; pad_id: token that takes position pad_id (-1: none); ids outside [0, vocab) read row 0

// out[u,:] = src[u/up,:] + (src[u/up,0] != pad_value ? pos_row : 0)   (CTC unit decoder input,
// reference ctc_unity/modules/ctc_transformer_unit_decoder.py:153-181)
int launch_upsample_add_pos(const float* src, int n, int up, const float* pos_row, float pad_value,
                            float* out, int D, hipStream_t stream);

// ids[m] = argmax_n logits[m,n] over n not in {mask0,mask1,mask2} (first max wins);
// if force >= 0 the result is `force` (beam-search max-length rule).  Optionally max value out.
int launch_masked_argmax(const float* logits, int ld, int M, int N, int mask0, int mask1, int mask2,
                         int force, int* ids, hipStream_t stream, const int* row_max_len = nullptr, int step = 0,
                         int force_id = -1);  // row_max_len: force `force_id` on rows with step >= row_max_len[row]

// out[m] = max_{n not in masks} log_softmax(logits[m,:])[n]   (researches/ctc_unity/ctc_generator.py:55-63)
int launch_log_softmax(const float* logits, int ld, int M, int N, int mask0, int mask1, int as_probs, float* out, int ldo,
                       hipStream_t stream);
int launch_row_max_logprob(const float* logits, int ld, int M, int N, int mask0, int mask1, int mask2, float* out,
                           hipStream_t stream);

// CTC collapse (reference agent/ctc_decoder.py:66-88): drop repeats, then drop `blank` and `pad`.
// tokens/index get the survivors and their frame index; *count their number.  Single workgroup.
int launch_ctc_collapse(const int* raw, int T, int blank, int pad, int* tokens, int* index, int* count,
                        hipStream_t stream, const int* segs = nullptr, int nseg = 0);  // segs {start,len}: count[s]

// emb_out[k,:] = table[codes[k],:]
int launch_gather_rows(const int* idx, const float* table, int D, float* out, int n, hipStream_t stream, int rows);  // ids outside [0, rows) read row 0

// dur[k] = clamp(round_half_even(exp(logdur[k]) - 1), min 1)   (reference agent/tts/codehifigan.py:61-64);
// forced != null overrides the prediction.  cum[0..K] = exclusive prefix sum (cum[K] = total frames).
// Single workgroup.
int launch_dur_predict(const float* logdur, const int* forced, int K, int* dur, int* cum, hipStream_t stream,
                       const int* segs = nullptr, int nseg = 0);  // segs {start,len}; cum of s at start+s

// out[f,:] = emb[k(f),:], k(f) = the unit whose [cum[k], cum[k+1]) holds f  (torch.repeat_interleave)
int launch_repeat_rows(const float* emb, const int* cum, int K, int D, float* out, int F, hipStream_t stream,
                       const int* segs = nullptr, int nseg = 0);  // segs {unit_start,n_units,frame_start,n_frames}; F = max

// wav[t] = tanh(b + sum_{j<7,c<C} w[j*C+c] * lrelu(x[t+j-3, c], slope))   (HiFi-GAN conv_post,
// reference fairseq/models/text_to_speech/hifigan.py:166-168; slope = 0.01)
int launch_conv_post_tanh(const float* x, int T, int C, const float* w, const float* bias, float slope,
                          float* wav, hipStream_t stream, const int* segs = nullptr, int nseg = 0);  // {start,len}; T = max

}  // namespace ss
