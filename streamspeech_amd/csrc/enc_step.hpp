// Persistent multi-phase launches of one Conformer layer of the incremental streaming encoder (enc_step.hip).
#pragma once
#include "common.hpp"

namespace ss {

constexpr int ES_D = 256;        // encoder width
constexpr int ES_F = 2048;       // FFN width
constexpr int ES_MAXR = 48;      // rows (non-final frames of a streaming call) a launch takes
constexpr int ES_G = 64;         // resident workgroups of a launch

struct EsLayerW {
  const float *ffn1_ln_g, *ffn1_ln_b, *ffn1_w1, *ffn1_b1, *ffn1_w2, *ffn1_b2;
  const float *attn_ln_g, *attn_ln_b, *qkv_w, *qkv_b, *out_w, *out_b;
  const float *conv_ln_g, *conv_ln_b, *pw1_w, *dw_wt, *bn_mean, *bn_var, *bn_g, *bn_b, *pw2_w;
  const float *ffn2_ln_g, *ffn2_ln_b, *ffn2_w1, *ffn2_b1, *ffn2_w2, *ffn2_b2, *final_ln_g, *final_ln_b;
};

struct EsArgs {
  EsLayerW w;
  float* x;            // [n][256] running activations of the rows being (re)computed
  float* qkv;          // layer cache [cap][768], absolute rows (rows r0 .. r0 + n - 1 are written)
  float* glu;          // layer cache [cap][256], absolute rows
  float* hctx;         // [n][256] attention context (written by the attention launch between the two persistent launches)
  float* g2;                  // [48][256] depthwise-conv output (scratch of the launch)
  float* part;                // [ES_G][ES_MAXR][256] partial FFN outputs
  unsigned* bar;              // monotone arrival counter of this scratch set (zeroed once)
  unsigned bar_base;          // its value when this launch starts
  unsigned* err;              // bounded-wait time-outs (must stay 0): the device word every waiting workgroup polls
  unsigned* err_host;         // the same count in pinned host memory, written only BY a time-out (read by the host after its synchronisation)
  int n, r0, T2, cchunk, dwk;
  int ph0, ph1;        // phases [ph0, ph1] of the layer (0-2: FFN1 + QKV; 4-9: attention output ... FFN2 + final LayerNorm)
};

size_t enc_step_lds_bytes();
size_t enc_step_scratch_bytes();
int launch_enc_step(const EsArgs& a, hipStream_t stream);
long long enc_step_launch_count();     // process-wide, for tests

}  // namespace ss
