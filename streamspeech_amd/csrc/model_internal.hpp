// Internals shared by the stage / batch / vocoder / debug translation units of the C ABI (include/streamspeech_hip.h): the context
// structs (ss_model, ss_vocoder: weights by slot name, scratch, caches, per-context settings) and the small host helpers every stage
// uses.  Round 5 split the 1 900-line model.hip into
//   model.hip     context creation / weights, front-end, single-utterance stages (encoder, streaming encoder, CTC, MT, T2U)
//   batch.hip     the ragged-batch twins of those stages (ss_batch_*)
//   vocoder.hip   the unit HiFi-GAN: context, generator stack, single-utterance and ragged-batch forward
//   debug_ops.hip op-level unit-test entry points (ss_op_*), test hooks (ss_debug_*), the profiler's C ABI (ss_prof_*)
#pragma once
#include <algorithm>
#include <atomic>
#include <map>
#include <mutex>
#include <string>
#include <tuple>
#include <unordered_map>
#include <vector>

#include "../../include/streamspeech_hip.h"
#include "attention.hpp"
#include "common.hpp"
#include "elementwise.hpp"
#include "fbank.hpp"
#include "gemm.hpp"
#include "mt_step.hpp"
#include "enc_step.hpp"

using namespace ss;

namespace {

struct Slot { const float* p = nullptr; int64_t n = 0; };

struct WeightTable {
  std::unordered_map<std::string, Slot> map;
  std::string missing;
  int build(const float* blob, size_t blob_floats, const char* const* names, const int64_t* offs,
            const int64_t* numels, int n) {
    for (int i = 0; i < n; ++i) {
      if (offs[i] < 0 || (size_t)(offs[i] + numels[i]) > blob_floats) return SS_ERR_ARG;
      map[names[i]] = Slot{blob + offs[i], numels[i]};
    }
    return SS_OK;
  }
  const float* get(const std::string& name, int64_t expect) {
    auto it = map.find(name);
    if (it == map.end() || (expect > 0 && it->second.n != expect)) {
      if (missing.empty()) {
        missing = name;
        fprintf(stderr, "[streamspeech_hip] weight slot '%s' missing or wrong size (want %lld, have %lld)\n",
                name.c_str(), (long long)expect, it == map.end() ? -1LL : (long long)it->second.n);
      }
      return nullptr;
    }
    return it->second.p;
  }
};

// Bytes held by the buffers of one scratch set (ss_scratch): `cap` > 0 bounds them (ss_scratch_set_cap)
struct ScratchAcct { size_t cap = 0, used = 0; };

struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
  ScratchAcct* acct = nullptr;       // the scratch set this buffer is booked under (null: a weight-side buffer, not capped)
  int ensure(size_t need) {
    if (need <= bytes) return SS_OK;
    size_t cap = need + std::min(need / 4, (size_t)256 << 20) + 4096;   // growth slack: a quarter, at most 256 MB (a pack's activations run to 12 GB)
    if (acct && acct->cap) {
      const size_t others = acct->used - bytes;
      if (others + cap > acct->cap) cap = need;                          // no slack under a cap ...
      if (others + cap > acct->cap) return SS_ERR_SCRATCH_CAP;           // ... and a clean error past it (nothing was freed)
    }
    if (p) { SS_HIP_CHECK(hipDeviceSynchronize()); SS_HIP_CHECK(hipFree(p)); p = nullptr; if (acct) acct->used -= bytes; bytes = 0; }
    SS_HIP_CHECK(hipMalloc(&p, cap));
    bytes = cap;
    if (acct) acct->used += cap;
    return SS_OK;
  }
  void release() { if (p) { (void)hipFree(p); p = nullptr; if (acct) acct->used -= bytes; bytes = 0; } }
  float* f() const { return reinterpret_cast<float*>(p); }
};

struct Lin { const float* w = nullptr; const float* b = nullptr; };
struct LN { const float* g = nullptr; const float* b = nullptr; };

#define RET(x) do { int _r = (x); if (_r != SS_OK) return _r; } while (0)
static int mt_persistent_env() {
  const char* e = getenv("SS_MT_PERSISTENT");
  const int v = e ? atoi(e) : 0;
  return (v == 64 || v == 128 || v == 256) ? v : 0;       // anything else: the launch-per-op form
}
static const int g_mt_persistent_default = mt_persistent_env();   // default of ss_mt_set_persistent for new contexts (0: launch-per-op decode step)
static const int g_no_mt_device_loop = getenv("SS_NO_MT_DEVICE_LOOP") ? atoi(getenv("SS_NO_MT_DEVICE_LOOP")) : 0;   // A/B knob: one persistent launch per TOKEN (round 3) instead of one per search
static const int g_no_enc_step = getenv("SS_NO_ENC_STEP") ? atoi(getenv("SS_NO_ENC_STEP")) : 0;   // A/B knob: the streaming encoder's layers as one launch per op even on a persistent context
static const int g_no_mt_ln_fusion = getenv("SS_NO_MT_LN_FUSION") ? atoi(getenv("SS_NO_MT_LN_FUSION")) : 0;   // A/B knob: separate final LayerNorm launch in the MT decode step

[[maybe_unused]] int linear(hipStream_t s, const float* A, int lda, int M, const Lin& l, int N, int K, float* C, int ldc,
           int act = ACT_NONE, float alpha = 1.f, const float* R = nullptr, int ldr = 0, int glu = 0) {
  GemmArgs a;
  a.A = A; a.lda = lda; a.W = l.w; a.bias = l.b; a.C = C; a.ldc = ldc; a.R = R; a.ldr = ldr;
  a.M = M; a.N = N; a.Cin = K; a.in_len = M; a.act = act; a.alpha = alpha; a.glu = glu;
  a.same_rows = 1;   // a linear layer maps row m to row m
  return launch_conv_gemm(a, s);
}

[[maybe_unused]] int layernorm(hipStream_t s, const float* x, float* y, const LN& ln, int M, int D) {
  return launch_layernorm(x, D, y, D, ln.g, ln.b, M, D, 1e-5f, s);
}

// C = epilogue(LayerNorm(x) @ W^T): fused into the GEMM prologue on the small-M path, otherwise a
// LayerNorm launch into `h` followed by the GEMM.
[[maybe_unused]] int ln_linear(hipStream_t s, const float* x, int M, const LN& ln, const Lin& l, int N, int K, float* C, int ldc,
              float* h, int act = ACT_NONE, float alpha = 1.f, int glu = 0) {
  GemmArgs a;
  a.A = x; a.lda = K; a.W = l.w; a.bias = l.b; a.C = C; a.ldc = ldc;
  a.M = M; a.N = N; a.Cin = K; a.in_len = M; a.act = act; a.alpha = alpha; a.glu = glu;
  a.same_rows = 1;
  const int canon = debug_tile_forced() ? CANON_NONE : canon_mode();
  if (canon == CANON_SMALLM) {                // lock-step MT decode rows: LayerNorm in the small-M kernel's prologue, always
    a.ln_g = ln.g; a.ln_b = ln.b;
    return launch_conv_gemm(a, s);
  }
  if (canon == CANON_SEQ) {
    // pack-invariant: the LayerNorm form must not change with the row count -- K = 256 layers always take the row-tile kernel's
    // (rtlin.hip), the others always the LayerNorm kernel + a GEMM
    GemmArgs b = a;
    b.ln_g = ln.g; b.ln_b = ln.b;
    if (rtlin_shape_ok(b)) return launch_conv_gemm(b, s);
    int rc = layernorm(s, x, h, ln, M, K);
    if (rc != SS_OK) return rc;
    a.A = h;
    return launch_conv_gemm(a, s);
  }
  if (smallm_eligible(a) && K <= 512) {
    a.ln_g = ln.g; a.ln_b = ln.b;
    return launch_conv_gemm(a, s);
  }
  if (!debug_tile_forced()) {                 // (a forced tile keeps launch_conv_gemm off the row-tile kernel: ADVICE r4)
    GemmArgs b = a;
    b.ln_g = ln.g; b.ln_b = ln.b;
    if (rtlin_eligible(b)) return launch_conv_gemm(b, s);      // LayerNorm in the row tile's way into LDS (rtlin.hip)
  }
  int rc = layernorm(s, x, h, ln, M, K);
  if (rc != SS_OK) return rc;
  a.A = h;
  return launch_conv_gemm(a, s);
}

}  // namespace

static const int g_pack_invariant_default = getenv("SS_PACK_INVARIANT") ? atoi(getenv("SS_PACK_INVARIANT")) : 1;   // A/B knob: default of ss_model_set_pack_invariant for new contexts

// =================================================================================================
// model
// =================================================================================================
struct EncLayer {
  LN ffn1_ln, attn_ln, conv_ln, ffn2_ln, final_ln;
  Lin ffn1_w1, ffn1_w2, ffn2_w1, ffn2_w2, qkv, out, pw1, pw2;
  const float *u, *v, *dw_wt, *bn_mean, *bn_var, *bn_g, *bn_b;
};
struct DecLayer {
  LN self_ln, cross_ln, ffn_ln;
  Lin self_qkv, self_out, cross_q, cross_kv, cross_out, fc1, fc2;
  bool has_cross = false;
};

// Everything a call MUTATES: activations, KV caches, the stream-K hand-off state, streaming-encoder state, the token chain.
// One per concurrent stream (ss_scratch_create); any number of weight handles (ss_model / ss_vocoder: one per language, cheap) may be
// bound to it one after the other (ss_model_bind_scratch / ss_vocoder_bind_scratch) -- round 5 welded a scratch set into every weight
// handle, so L languages on S streams cost L x S scratch sets.  Ref-counted: freed when the creator and
// every bound handle have let go.
struct ss_scratch {
  std::atomic<int> refs{1};
  ScratchAcct acct;
  SkWorkspace* skws = nullptr;       // stream-K hand-off state of this context
  // ---- model side ----
  DevBuf ws;            // encoder / t2u / unit scratch
  DevBuf mt_cross;      // [mt_layers][Tp][2*D]
  DevBuf mt_self;       // [mt_layers][max_tgt_pos][3*D]
  DevBuf mt_ws;         // per-append scratch
  int mt_Tp = 0;
  int mt_len = 0;
  const float* mt_enc = nullptr;
  DevBuf attn_split;             // key-split scratch of the single-utterance rel-pos attention (attention.hpp); counters zeroed once
  DevBuf mt_gran;                // persistent decode step (mt_step.hip): granule region, zeroed once; the epoch grows per launch
  unsigned mt_epoch = 0;
  hipStream_t mt_last_stream = nullptr;   // stream of the last persistent-step launch (mt_collect_errors reads / clears the error word there)
  int mt_inject_timeout = 0;     // ss_debug_mt_inject_timeout: the next persistent launch reports a time-out
  int mt_persistent = g_mt_persistent_default;   // workgroups of the persistent decode step (ss_mt_set_persistent); 0 = launch-per-op
  DevBuf mt_tok;                 // device token chain [max_tgt_pos] (greedy search feeds itself)
  DevBuf seg_buf;                // ragged-batch segment tables / batched token chain
  DevBuf bmt_self;               // batched MT self-attention cache [layer][B][Lcap][3D]
  int32_t* mt_tok_host = nullptr;  // pinned staging of the same
  size_t mt_tok_host_n = 0;
  // incremental streaming encoder (ss_encoder_stream_*): per-layer fused q|k|v rows and GLU outputs
  // of every frame so far + the finished output rows; rows < es_final are final
  DevBuf es_qkv;        // [layers][es_cap][3d]
  DevBuf es_glu;        // [layers][es_cap][d]
  DevBuf es_out;        // [es_cap][d]
  DevBuf es_step;       // persistent layer launches (enc_step.hip): partial FFN outputs, depthwise output, arrival counter + error word (zeroed once)
  unsigned es_bar = 0;  // value of the arrival counter when the next launch starts
  int es_step_off = 0;  // 1 after a time-out: this scratch set stays on one launch per op
  int es_deferred = 0;  // ss_encoder_stream_set_deferred: the persistent form's time-out check is left to ss_encoder_stream_status
  int es_pending = 0;   // a deferred check is outstanding (the last forward's output must not be trusted yet)
  int es_final_prev = 0;   // es_final before that forward (restored when the check fails)
  int es_inject = 0;    // test hook: the next check reports a time-out
  unsigned* es_err_host = nullptr;   // pinned host word the persistent layer launches bump on a time-out (next to the device word the workgroups poll):
                                     // the host reads it after its stream synchronisation -- no device-to-host copy per call
  int es_cap = 0, es_final = 0, es_achunk = -1, es_cchunk = -1;
  int es_tail = 0;                                  // trailing fbank frames that may still change (resampler edge)
  // ss_debug_last_logits: where the last batched argmax stage of this context left its dense logits (scratch, valid until the
  // next call that uses the same scratch buffer)
  const float* dbg_logits = nullptr;
  int dbg_rows = 0, dbg_cols = 0;
  // ---- vocoder side ----
  DevBuf v_ws, v_small, v_segs;
  ss_scratch() {
    for (DevBuf* b : all()) b->acct = &acct;
    skws = sk_workspace_new();
  }
  ~ss_scratch() {
    for (DevBuf* b : all()) b->release();
    if (mt_tok_host) (void)hipHostFree(mt_tok_host);
    if (es_err_host) (void)hipHostFree(es_err_host);
    sk_workspace_free(skws);
  }
  std::vector<DevBuf*> all() {
    return {&ws, &mt_cross, &mt_self, &mt_ws, &attn_split, &mt_gran, &mt_tok, &seg_buf, &bmt_self, &es_qkv, &es_glu, &es_out, &es_step, &v_ws, &v_small, &v_segs};
  }
  // what ss_scratch_trim may let go: buffers every entry point re-sizes before use (the zero-initialised ones and the KV cache stay)
  std::vector<DevBuf*> trimmable() { return {&ws, &mt_cross, &mt_ws, &seg_buf, &bmt_self, &es_qkv, &es_glu, &es_out, &v_ws, &v_small, &v_segs}; }
};
[[maybe_unused]] static void scratch_unref(ss_scratch* sc) { if (sc && sc->refs.fetch_sub(1) == 1) delete sc; }
// the fixed-size pieces a model of configuration `c` needs in the scratch set it runs on: the MT self-attention cache, the token chain
[[maybe_unused]] static int scratch_fit_model(ss_scratch* sc, const ss_config& c) {
  int rc = sc->mt_self.ensure((size_t)c.mt_layers * c.max_tgt_pos * 3 * c.dec_dim * sizeof(float));
  if (rc == SS_OK) rc = sc->mt_tok.ensure((size_t)c.max_tgt_pos * sizeof(int32_t));
  if (rc == SS_OK && sc->mt_tok_host_n < (size_t)c.max_tgt_pos) {
    if (sc->mt_tok_host) (void)hipHostFree(sc->mt_tok_host);
    sc->mt_tok_host = nullptr; sc->mt_tok_host_n = 0;
    if (hipHostMalloc((void**)&sc->mt_tok_host, (size_t)c.max_tgt_pos * sizeof(int32_t)) != hipSuccess) return SS_ERR_HIP;
    sc->mt_tok_host_n = (size_t)c.max_tgt_pos;
  }
  return rc;
}

struct ss_model {
  ss_config cfg;
  WeightTable wt;
  ss_scratch* sc = nullptr;          // the scratch set this handle is bound to (its own unless ss_model_bind_scratch said otherwise)
  // encoder
  Lin sub0, sub1, enc_linear, ctc_asr, ctc_st;
  std::vector<EncLayer> enc;
  const float* pos_table = nullptr;  // [2*Tmax-1, d]
  const float* pos_w = nullptr;      // [L*d, d]
  const float* pos_proj = nullptr;   // [2*Tmax-1, L*d]: a function of the blob -- ONE buffer per blob and device, shared by every handle over it (model.hip)
  int pos_key_dev = 0;
  // front-end
  const float *fe_window = nullptr, *fe_melw = nullptr, *fe_mean = nullptr, *fe_std = nullptr;
  // decoders
  const float* mt_emb = nullptr; const float* mt_pos = nullptr; LN mt_ln;
  std::vector<DecLayer> mt, t2u, unit;
  LN t2u_ln, unit_ln;
  Lin unit_out;
  const float* unit_pos_row = nullptr;
  // ss_model_set_pack_invariant: 1 = every ss_batch_* stage upstream of an arg-max computes a packed utterance with arithmetic that
  // is a function of that utterance alone (same bits alone, in any pack, at any position); 0 = fastest kernel per shape (round-4 routes)
  int pack_invariant = g_pack_invariant_default;
};

// Key-split scratch of this context for the single-utterance rel-pos attention: allocated and zeroed on first use (the
// counters must read zero; the stream is synchronised once so that a later call on another stream sees them).
[[maybe_unused]] static int bind_attn_split(ss_model* m, AttnArgs& at, hipStream_t s) {
  if (!m->sc->attn_split.p) {
    RET(m->sc->attn_split.ensure(attention_split_bytes()));
    SS_HIP_CHECK(hipMemsetAsync(m->sc->attn_split.p, 0, attention_split_bytes(), s));
    SS_HIP_CHECK(hipStreamSynchronize(s));
  }
  attention_bind_split(at, m->sc->attn_split.p);
  return SS_OK;
}

[[maybe_unused]] static int conv_out_len(int L, int k, int stride) { return (L + 2 * (k / 2) - k) / stride + 1; }

// ---- transformer layers shared by MT decoder / T2U encoder / unit decoder ----------------------
// x [n, D] in place.  self K/V cache rows live in `selfbuf` ([*, 3D], row = absolute position).
// One pre-LN transformer layer on the residual stream x [n, D].  The fused QKV rows are written to
// `qkv_rows` with row stride ld_qkv (straight into a KV cache when decoding); the caller prepares
// the attention descriptors (single utterance or ragged batch) -- their O is `h`, cross Q is `q2`.
[[maybe_unused]] static int dec_layer_ex(hipStream_t s, const ss_config& c, const DecLayer& L, float* x, int n, float* qkv_rows,
                        int ld_qkv, const AttnArgs& self_at, const AttnArgs* cross_at, float* h, float* q2, float* ff) {
  const int D = c.dec_dim, F = c.dec_ffn;
  RET(ln_linear(s, x, n, L.self_ln, L.self_qkv, 3 * D, D, qkv_rows, ld_qkv, h));   // q (pre-scaled at pack time), k, v
  RET(launch_attention(self_at, s));
  RET(linear(s, h, D, n, L.self_out, D, D, x, D, ACT_NONE, 1.f, x, D));
  if (L.has_cross && cross_at) {
    RET(ln_linear(s, x, n, L.cross_ln, L.cross_q, D, D, q2, D, h));
    RET(launch_attention(*cross_at, s));
    RET(linear(s, h, D, n, L.cross_out, D, D, x, D, ACT_NONE, 1.f, x, D));
  }
  RET(ln_linear(s, x, n, L.ffn_ln, L.fc1, F, D, ff, F, h, ACT_RELU));
  RET(linear(s, ff, F, n, L.fc2, D, F, x, D, ACT_NONE, 1.f, x, D));
  return SS_OK;
}

// single utterance: self K/V cache rows live in `selfbuf` ([*, 3D], row = absolute position)
[[maybe_unused]] static int dec_layer(hipStream_t s, const ss_config& c, const DecLayer& L, float* x, int n, int pos0,
                     float* selfbuf, bool causal, const float* crossKV, int Tk_cross, float* h, float* q2,
                     float* ff, int self_tail_pad = 0, int cross_tail_pad = 0) {
  const int D = c.dec_dim, H = c.dec_heads;
  float* rows = selfbuf + (size_t)pos0 * 3 * D;
  AttnArgs at;
  at.Q = rows; at.ldq = 3 * D; at.K = selfbuf + D; at.V = selfbuf + 2 * D; at.ldk = at.ldv = 3 * D;
  at.O = h; at.ldo = D; at.Tq = n; at.Tk = pos0 + n; at.H = H; at.scale = 1.f; at.causal = causal ? 1 : 0;
  at.k_mask_tail = self_tail_pad;
  AttnArgs ac;
  if (L.has_cross) {
    ac.Q = q2; ac.ldq = D; ac.K = crossKV; ac.V = crossKV + D; ac.ldk = ac.ldv = 2 * D;
    ac.O = h; ac.ldo = D; ac.Tq = n; ac.Tk = Tk_cross; ac.H = H; ac.scale = 1.f; ac.k_mask_tail = cross_tail_pad;
  }
  return dec_layer_ex(s, c, L, x, n, rows, 3 * D, at, L.has_cross ? &ac : nullptr, h, q2, ff);
}

extern std::atomic<int> g_mt_timeouts;      // bounded-wait time-outs of persistent MT decode steps, process-wide (model.hip)

struct ConvW { const float* w = nullptr; const float* b = nullptr; const float* ww = nullptr; };   // ww: Winograd form (64-channel stage ResBlock convs)
struct ss_vocoder {
  ss_vocoder_config cfg;
  WeightTable wt;
  ss_scratch* sc = nullptr;          // the scratch set this handle is bound to (ss_vocoder_bind_scratch; its own by default)
  const float* dict = nullptr;
  ConvW dur_c1, dur_c2, dur_proj, pre, post;
  LN dur_ln1, dur_ln2;
  std::vector<ConvW> ups;
  std::vector<ConvW> rb_c1, rb_c2;  // [(stage*n_res + j)*3 + d]
  float* wino = nullptr;             // Winograd F(2,3) forms of the 32- / 64- / 128-channel stages' ResBlock conv weights (conv_c64w.hip):
  const float* wino_key = nullptr;   // ONE buffer per weight blob, shared by every context over that blob (wino_share below)
  int x3 = 0;          // split-bf16 contraction of the C >= 64 generator convs (ss_vocoder_set_bf16x3); default off = exact f32
};

namespace {

// small int tables for the kernels: pageable -> device copies are staged by the runtime before
// hipMemcpyAsync returns, so the std::vector may die right after the call
[[maybe_unused]] int upload(hipStream_t s, int* dst, const std::vector<int>& v) {
  SS_HIP_CHECK(hipMemcpyAsync(dst, v.data(), v.size() * sizeof(int), hipMemcpyHostToDevice, s));
  return SS_OK;
}

struct Offsets { std::vector<int> off; int total = 0, mx = 0; };
[[maybe_unused]] Offsets prefix(const int* len, int B) {
  Offsets o; o.off.resize(B + 1); o.off[0] = 0;
  for (int b = 0; b < B; ++b) { o.off[b + 1] = o.off[b] + len[b]; o.mx = std::max(o.mx, len[b]); }
  o.total = o.off[B];
  return o;
}

}  // namespace
