// Stage orchestration + C ABI for the StreamSpeech S2ST path (include/streamspeech_hip.h).
// Host code only queues kernels on the caller's stream; the only device->host syncs are the
// frame count in the vocoder and whatever the caller does with the returned ids.
#include <algorithm>
#include <atomic>
#include <map>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/streamspeech_hip.h"
#include "attention.hpp"
#include "common.hpp"
#include "elementwise.hpp"
#include "fbank.hpp"
#include "gemm.hpp"
#include "mt_step.hpp"

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

struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
  int ensure(size_t need) {
    if (need <= bytes) return SS_OK;
    if (p) { SS_HIP_CHECK(hipDeviceSynchronize()); SS_HIP_CHECK(hipFree(p)); p = nullptr; bytes = 0; }
    size_t cap = need + need / 4 + 4096;
    SS_HIP_CHECK(hipMalloc(&p, cap));
    bytes = cap;
    return SS_OK;
  }
  void release() { if (p) { (void)hipFree(p); p = nullptr; bytes = 0; } }
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
static const int g_no_mt_ln_fusion = getenv("SS_NO_MT_LN_FUSION") ? atoi(getenv("SS_NO_MT_LN_FUSION")) : 0;   // A/B knob: separate final LayerNorm launch in the MT decode step

int linear(hipStream_t s, const float* A, int lda, int M, const Lin& l, int N, int K, float* C, int ldc,
           int act = ACT_NONE, float alpha = 1.f, const float* R = nullptr, int ldr = 0, int glu = 0) {
  GemmArgs a;
  a.A = A; a.lda = lda; a.W = l.w; a.bias = l.b; a.C = C; a.ldc = ldc; a.R = R; a.ldr = ldr;
  a.M = M; a.N = N; a.Cin = K; a.in_len = M; a.act = act; a.alpha = alpha; a.glu = glu;
  a.same_rows = 1;   // a linear layer maps row m to row m
  return launch_conv_gemm(a, s);
}

int layernorm(hipStream_t s, const float* x, float* y, const LN& ln, int M, int D) {
  return launch_layernorm(x, D, y, D, ln.g, ln.b, M, D, 1e-5f, s);
}

// C = epilogue(LayerNorm(x) @ W^T): fused into the GEMM prologue on the small-M path, otherwise a
// LayerNorm launch into `h` followed by the GEMM.
int ln_linear(hipStream_t s, const float* x, int M, const LN& ln, const Lin& l, int N, int K, float* C, int ldc,
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

struct ss_model {
  ss_config cfg;
  WeightTable wt;
  SkWorkspace* skws = nullptr;       // stream-K hand-off state of this context (freed with the handle)
  // encoder
  Lin sub0, sub1, enc_linear, ctc_asr, ctc_st;
  std::vector<EncLayer> enc;
  const float* pos_table = nullptr;  // [2*Tmax-1, d]
  const float* pos_w = nullptr;      // [L*d, d]
  DevBuf pos_proj;                   // [2*Tmax-1, L*d]
  // front-end
  const float *fe_window = nullptr, *fe_melw = nullptr, *fe_mean = nullptr, *fe_std = nullptr;
  // decoders
  const float* mt_emb = nullptr; const float* mt_pos = nullptr; LN mt_ln;
  std::vector<DecLayer> mt, t2u, unit;
  LN t2u_ln, unit_ln;
  Lin unit_out;
  const float* unit_pos_row = nullptr;
  // scratch
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
  // incremental streaming encoder (ss_encoder_stream_*): per-layer fused q|k|v rows and GLU outputs
  // of every frame so far + the finished output rows; rows < es_final are final
  DevBuf es_qkv;        // [layers][es_cap][3d]
  DevBuf es_glu;        // [layers][es_cap][d]
  DevBuf es_out;        // [es_cap][d]
  int es_cap = 0, es_final = 0, es_achunk = -1, es_cchunk = -1;
  int es_tail = 0;                                  // trailing fbank frames that may still change (resampler edge)
  // ss_debug_last_logits: where the last batched argmax stage of this context left its dense logits (scratch, valid until the
  // next call that uses the same scratch buffer)
  const float* dbg_logits = nullptr;
  int dbg_rows = 0, dbg_cols = 0;
  // ss_model_set_pack_invariant: 1 = every ss_batch_* stage upstream of an arg-max computes a packed utterance with arithmetic that
  // is a function of that utterance alone (same bits alone, in any pack, at any position); 0 = fastest kernel per shape (round-4 routes)
  int pack_invariant = g_pack_invariant_default;
};

// Key-split scratch of this context for the single-utterance rel-pos attention: allocated and zeroed on first use (the
// counters must read zero; the stream is synchronised once so that a later call on another stream sees them).
static int bind_attn_split(ss_model* m, AttnArgs& at, hipStream_t s) {
  if (!m->attn_split.p) {
    RET(m->attn_split.ensure(attention_split_bytes()));
    SS_HIP_CHECK(hipMemsetAsync(m->attn_split.p, 0, attention_split_bytes(), s));
    SS_HIP_CHECK(hipStreamSynchronize(s));
  }
  attention_bind_split(at, m->attn_split.p);
  return SS_OK;
}

static int load_dec_layers(ss_model* m, std::vector<DecLayer>& v, const std::string& pfx, int n, int D, int F,
                           int kv_in, bool cross) {
  WeightTable& w = m->wt;
  v.resize(n);
  for (int l = 0; l < n; ++l) {
    const std::string p = pfx + ".L" + std::to_string(l) + ".";
    DecLayer& d = v[l];
    d.has_cross = cross;
    d.self_ln = {w.get(p + "self.ln.g", D), w.get(p + "self.ln.b", D)};
    d.self_qkv = {w.get(p + "self.qkv.w", 3LL * D * D), w.get(p + "self.qkv.b", 3LL * D)};
    d.self_out = {w.get(p + "self.out.w", (int64_t)D * D), w.get(p + "self.out.b", D)};
    if (cross) {
      d.cross_ln = {w.get(p + "cross.ln.g", D), w.get(p + "cross.ln.b", D)};
      d.cross_q = {w.get(p + "cross.q.w", (int64_t)D * D), w.get(p + "cross.q.b", D)};
      d.cross_kv = {w.get(p + "cross.kv.w", 2LL * D * kv_in), w.get(p + "cross.kv.b", 2LL * D)};
      d.cross_out = {w.get(p + "cross.out.w", (int64_t)D * D), w.get(p + "cross.out.b", D)};
    }
    d.ffn_ln = {w.get(p + "ffn.ln.g", D), w.get(p + "ffn.ln.b", D)};
    d.fc1 = {w.get(p + "fc1.w", (int64_t)F * D), w.get(p + "fc1.b", F)};
    d.fc2 = {w.get(p + "fc2.w", (int64_t)D * F), w.get(p + "fc2.b", D)};
  }
  return SS_OK;
}

extern "C" int ss_abi_version(void) { return SS_ABI_VERSION; }

extern "C" const char* ss_error_string(int code) {
  switch (code) {
    case SS_OK: return "ok";
    case SS_ERR_HIP: return "HIP runtime error";
    case SS_ERR_ARG: return "invalid argument";
    case SS_ERR_MISSING_WEIGHT: return "weight slot missing or wrong size";
    case SS_ERR_CAPACITY: return "output capacity too small";
    default: return "unknown error";
  }
}

extern "C" int ss_model_create(const ss_config* cfg, const float* d_blob, size_t blob_floats,
                               const char* const* names, const int64_t* offsets, const int64_t* numels,
                               int n_slots, ss_model** out) {
  if (!cfg || !d_blob || !out) return SS_ERR_ARG;
  if (cfg->enc_dim / cfg->enc_heads != 64 || cfg->dec_dim / cfg->dec_heads != 64) return SS_ERR_ARG;
  ss_model* m = new ss_model();
  m->cfg = *cfg;
  m->skws = sk_workspace_new();
  SkScope sk_scope(m->skws);       // the pos_proj GEMM below may take a stream-K kernel
  int rc = m->wt.build(d_blob, blob_floats, names, offsets, numels, n_slots);
  if (rc != SS_OK) { sk_workspace_free(m->skws); delete m; return rc; }
  WeightTable& w = m->wt;
  const int d = cfg->enc_dim, f = cfg->enc_ffn, D = cfg->dec_dim, F = cfg->dec_ffn, k = cfg->conv_kernel;
  const int Tm = cfg->max_rel_pos;
  m->sub0 = {w.get("enc.sub0.w", (int64_t)cfg->conv_channels * k * cfg->input_feat), w.get("enc.sub0.b", cfg->conv_channels)};
  m->sub1 = {w.get("enc.sub1.w", (int64_t)2 * d * k * (cfg->conv_channels / 2)), w.get("enc.sub1.b", 2 * d)};
  m->enc_linear = {w.get("enc.linear.w", (int64_t)d * d), w.get("enc.linear.b", d)};
  m->pos_table = w.get("enc.pos_table", (int64_t)(2 * Tm - 1) * d);
  m->pos_w = w.get("enc.pos_w", (int64_t)cfg->enc_layers * d * d);
  m->enc.resize(cfg->enc_layers);
  for (int l = 0; l < cfg->enc_layers; ++l) {
    const std::string p = "enc.L" + std::to_string(l) + ".";
    EncLayer& e = m->enc[l];
    e.ffn1_ln = {w.get(p + "ffn1.ln.g", d), w.get(p + "ffn1.ln.b", d)};
    e.ffn1_w1 = {w.get(p + "ffn1.w1.w", (int64_t)f * d), w.get(p + "ffn1.w1.b", f)};
    e.ffn1_w2 = {w.get(p + "ffn1.w2.w", (int64_t)d * f), w.get(p + "ffn1.w2.b", d)};
    e.attn_ln = {w.get(p + "attn.ln.g", d), w.get(p + "attn.ln.b", d)};
    e.qkv = {w.get(p + "attn.qkv.w", 3LL * d * d), w.get(p + "attn.qkv.b", 3LL * d)};
    e.out = {w.get(p + "attn.out.w", (int64_t)d * d), w.get(p + "attn.out.b", d)};
    e.u = w.get(p + "attn.u", d);
    e.v = w.get(p + "attn.v", d);
    e.conv_ln = {w.get(p + "conv.ln.g", d), w.get(p + "conv.ln.b", d)};
    e.pw1 = {w.get(p + "conv.pw1.w", 2LL * d * d), nullptr};
    e.dw_wt = w.get(p + "conv.dw.wt", (int64_t)cfg->dw_kernel * d);
    e.bn_mean = w.get(p + "conv.bn.mean", d);
    e.bn_var = w.get(p + "conv.bn.var", d);
    e.bn_g = w.get(p + "conv.bn.g", d);
    e.bn_b = w.get(p + "conv.bn.b", d);
    e.pw2 = {w.get(p + "conv.pw2.w", (int64_t)d * d), nullptr};
    e.ffn2_ln = {w.get(p + "ffn2.ln.g", d), w.get(p + "ffn2.ln.b", d)};
    e.ffn2_w1 = {w.get(p + "ffn2.w1.w", (int64_t)f * d), w.get(p + "ffn2.w1.b", f)};
    e.ffn2_w2 = {w.get(p + "ffn2.w2.w", (int64_t)d * f), w.get(p + "ffn2.w2.b", d)};
    e.final_ln = {w.get(p + "final_ln.g", d), w.get(p + "final_ln.b", d)};
  }
  m->ctc_asr = {w.get("ctc.asr.w", (int64_t)cfg->src_vocab * d), w.get("ctc.asr.b", cfg->src_vocab)};
  m->ctc_st = {w.get("ctc.st.w", (int64_t)cfg->tgt_vocab * d), w.get("ctc.st.b", cfg->tgt_vocab)};
  m->fe_window = w.get("fe.window", 400);
  m->fe_melw = w.get("fe.melw", 80 * 257);
  m->fe_mean = w.get("fe.cmvn_mean", 80);
  m->fe_std = w.get("fe.cmvn_std", 80);
  m->mt_emb = w.get("mt.emb", (int64_t)cfg->tgt_vocab * D);
  m->mt_pos = w.get("mt.pos_table", (int64_t)cfg->max_tgt_pos * D);
  m->mt_ln = {w.get("mt.ln.g", D), w.get("mt.ln.b", D)};
  load_dec_layers(m, m->mt, "mt", cfg->mt_layers, D, F, d, true);
  load_dec_layers(m, m->t2u, "t2u", cfg->t2u_layers, D, F, D, false);
  m->t2u_ln = {w.get("t2u.ln.g", D), w.get("t2u.ln.b", D)};
  load_dec_layers(m, m->unit, "unit", cfg->unit_layers, D, F, D, true);
  m->unit_ln = {w.get("unit.ln.g", D), w.get("unit.ln.b", D)};
  m->unit_out = {w.get("unit.out.w", (int64_t)cfg->unit_vocab * D), nullptr};
  m->unit_pos_row = w.get("unit.pos_row", D);
  if (!w.missing.empty()) { sk_workspace_free(m->skws); delete m; return SS_ERR_MISSING_WEIGHT; }

  // projected rel-pos table for every layer at once: [2Tm-1, d] x [L*d, d]^T (linear_pos has no
  // bias, espnet_multihead_attention.py:125).  Depends only on the relative offset, so it is
  // computed once here and sliced per utterance.
  const int rows = 2 * Tm - 1, Ld = cfg->enc_layers * d;
  rc = m->pos_proj.ensure((size_t)rows * Ld * sizeof(float));
  if (rc == SS_OK) {
    Lin lp{m->pos_w, nullptr};
    rc = linear(nullptr, m->pos_table, d, rows, lp, Ld, d, m->pos_proj.f(), Ld);
  }
  if (rc == SS_OK && hipDeviceSynchronize() != hipSuccess) rc = SS_ERR_HIP;
  if (rc == SS_OK) rc = m->mt_self.ensure((size_t)cfg->mt_layers * cfg->max_tgt_pos * 3 * D * sizeof(float));
  if (rc == SS_OK) rc = m->mt_tok.ensure((size_t)cfg->max_tgt_pos * sizeof(int32_t));
  if (rc == SS_OK && hipHostMalloc((void**)&m->mt_tok_host, (size_t)cfg->max_tgt_pos * sizeof(int32_t)) != hipSuccess)
    rc = SS_ERR_HIP;
  if (rc != SS_OK) { ss_model_destroy(m); return rc; }
  *out = m;
  return SS_OK;
}

extern "C" void ss_model_destroy(ss_model* m) {
  if (!m) return;
  m->pos_proj.release(); m->ws.release(); m->mt_cross.release(); m->mt_self.release(); m->mt_ws.release();
  m->mt_tok.release(); m->seg_buf.release(); m->bmt_self.release(); m->mt_gran.release(); m->attn_split.release();
  m->es_qkv.release(); m->es_glu.release(); m->es_out.release();
  if (m->mt_tok_host) (void)hipHostFree(m->mt_tok_host);
  sk_workspace_free(m->skws);
  delete m;
}

// ---- front-end ---------------------------------------------------------------------------------
extern "C" int ss_resample(void* stream, const float* d_in, int64_t n_in, int up, int down, const float* d_taps,
                           int half_len, float* d_out, int64_t n_out) {
  if (!d_in || !d_out || !d_taps) return SS_ERR_ARG;
  return launch_resample(d_in, n_in, up, down, d_taps, half_len, d_out, n_out, (hipStream_t)stream);
}

extern "C" int ss_row_max_logprob(void* stream, const float* d_logits, int rows, int vocab, int mask0, int mask1, int mask2,
                                  float* d_out) {
  if (!d_logits || !d_out || rows < 0 || vocab <= 0) return SS_ERR_ARG;
  return launch_row_max_logprob(d_logits, vocab, rows, vocab, mask0, mask1, mask2, d_out, (hipStream_t)stream);
}

extern "C" int ss_log_softmax(void* stream, const float* d_logits, int rows, int vocab, int mask0, int mask1, int as_probs,
                              float* d_out) {
  if (!d_logits || !d_out || rows < 0 || vocab <= 0) return SS_ERR_ARG;
  return launch_log_softmax(d_logits, vocab, rows, vocab, mask0, mask1, as_probs, d_out, vocab, (hipStream_t)stream);
}

extern "C" int ss_fbank_num_frames(int n) { return n < 400 ? 0 : 1 + (n - 400) / 160; }

extern "C" int ss_fbank_cmvn(ss_model* m, void* stream, const float* d_pcm, int n_samples, float pcm_scale,
                             float* d_feat, int* h_n_frames) {
  if (!m) return SS_ERR_ARG;
  return launch_fbank_cmvn(d_pcm, n_samples, pcm_scale, m->fe_window, m->fe_melw, m->fe_mean, m->fe_std,
                           d_feat, h_n_frames, (hipStream_t)stream);
}

// ---- encoder -----------------------------------------------------------------------------------
static int conv_out_len(int L, int k, int stride) { return (L + 2 * (k / 2) - k) / stride + 1; }

extern "C" int ss_encoder_out_len(int T) {
  const int t1 = conv_out_len(T, 5, 2);
  return conv_out_len(t1, 5, 2);
}

extern "C" int ss_encoder_forward(ss_model* m, void* stream, const float* d_fbank, int T, int attn_chunk,
                                  int conv_chunk, float* d_enc_out) {
  if (!m || T <= 0) return SS_ERR_ARG;
  SkScope sk_scope(m->skws);
  hipStream_t s = (hipStream_t)stream;
  const ss_config& c = m->cfg;
  const int d = c.enc_dim, f = c.enc_ffn, k = c.conv_kernel, Ld = c.enc_layers * d;
  const int T1 = conv_out_len(T, k, 2), T2 = conv_out_len(T1, k, 2);
  if (T2 <= 0 || T2 > c.max_rel_pos) return SS_ERR_CAPACITY;
  const int cchunk = (conv_chunk > 0 && conv_chunk < 999) ? conv_chunk : 0;   // chunk_causal_conv1d.py:40
  const int achunk = (attn_chunk > 0 && attn_chunk < T2) ? attn_chunk : 0;

  // scratch layout (floats)
  const size_t n_h1 = (size_t)T1 * (c.conv_channels / 2);
  const size_t n_x = (size_t)T2 * d, n_f = (size_t)T2 * f, n_qkv = (size_t)T2 * 3 * d;
  RET(m->ws.ensure((n_h1 + 3 * n_x + n_f + n_qkv) * sizeof(float)));
  float* h1 = m->ws.f();
  float* x = d_enc_out;                 // running activations live in the output buffer
  float* h = h1 + n_h1;                 // LN output / attention context
  float* g = h + n_x;                   // GLU output / misc
  float* g2 = g + n_x;                  // depthwise output
  float* ff = g2 + n_x;                 // FFN hidden
  float* qkv = ff + n_f;

  // Conv1dSubsampler: (stride-2 k5 chunk-causal conv -> GLU) x 2   (convolution.py:81-89)
  {
    GemmArgs a;
    a.A = d_fbank; a.lda = c.input_feat; a.W = m->sub0.w; a.bias = m->sub0.b; a.C = h1; a.ldc = c.conv_channels / 2;
    a.M = T1; a.N = c.conv_channels; a.Cin = c.input_feat; a.taps = k; a.stride = 2; a.pad = k / 2;
    a.in_len = T; a.chunk = cchunk; a.glu = 1;
    RET(launch_conv_gemm(a, s));
    GemmArgs b;
    b.A = h1; b.lda = c.conv_channels / 2; b.W = m->sub1.w; b.bias = m->sub1.b; b.C = g; b.ldc = d;
    b.M = T2; b.N = 2 * d; b.Cin = c.conv_channels / 2; b.taps = k; b.stride = 2; b.pad = k / 2;
    b.in_len = T1; b.chunk = cchunk; b.glu = 1;
    RET(launch_conv_gemm(b, s));
  }
  // x = Linear(sqrt(d) * x)  -- the sqrt(d)=16 scale is folded (exactly) into enc.linear.w
  RET(linear(s, g, d, T2, m->enc_linear, d, d, x, d));
  const float* P = m->pos_proj.f() + (size_t)(c.max_rel_pos - T2) * Ld;

  for (int l = 0; l < c.enc_layers; ++l) {
    const EncLayer& e = m->enc[l];
    // x = x + 0.5 * FFN1(x)
    RET(ln_linear(s, x, T2, e.ffn1_ln, e.ffn1_w1, f, d, ff, f, h, ACT_SILU));
    RET(linear(s, ff, f, T2, e.ffn1_w2, d, f, x, d, ACT_NONE, 0.5f, x, d));
    // x = x + RelPosMHA(LN(x))
    RET(ln_linear(s, x, T2, e.attn_ln, e.qkv, 3 * d, d, qkv, 3 * d, h));
    AttnArgs at;
    at.Q = qkv; at.K = qkv + d; at.V = qkv + 2 * d; at.ldq = at.ldk = at.ldv = 3 * d;
    at.O = h; at.ldo = d; at.Tq = T2; at.Tk = T2; at.H = c.enc_heads; at.scale = 0.125f;
    at.chunk = achunk; at.P = P + (size_t)l * d; at.ldp = Ld; at.bias_u = e.u; at.bias_v = e.v;
    RET(bind_attn_split(m, at, s));
    RET(launch_attention(at, s));
    RET(linear(s, h, d, T2, e.out, d, d, x, d, ACT_NONE, 1.f, x, d));
    // x = x + ConvModule(x)
    RET(ln_linear(s, x, T2, e.conv_ln, e.pw1, 2 * d, d, g, d, h, ACT_NONE, 1.f, 1));
    RET(launch_dwconv_bn_silu(g, d, g2, d, e.dw_wt, c.dw_kernel, e.bn_mean, e.bn_var, e.bn_g, e.bn_b, 1e-5f,
                              T2, d, cchunk, s));
    RET(linear(s, g2, d, T2, e.pw2, d, d, x, d, ACT_NONE, 1.f, x, d));
    // x = LN(x + 0.5 * FFN2(x))
    RET(ln_linear(s, x, T2, e.ffn2_ln, e.ffn2_w1, f, d, ff, f, h, ACT_SILU));
    RET(linear(s, ff, f, T2, e.ffn2_w2, d, f, x, d, ACT_NONE, 0.5f, x, d));
    RET(layernorm(s, x, x, e.final_ln, T2, d));
  }
  return SS_OK;
}


// ---- incremental streaming encoder -------------------------------------------
// The reference re-runs the whole encoder on all audio received so far at every policy() call
// (agent/speech_to_speech.streamspeech.agent.py:425-435).  With chunk attention and chunk-causal
// convs a frame cannot see anything beyond the end of its chunk, so once the input that a chunk
// can reach has arrived its rows are FINAL: later calls reproduce them exactly.  This entry point
// keeps, per layer, the q|k|v rows and the conv-module GLU rows of every frame (the only things a
// later frame reads from an earlier one) plus the finished output rows, and runs the 12 conformer
// layers only on the rows that are not final yet.  The subsampler (6 % of the encoder FLOPs) is
// re-run in full: its chunk grid is in fbank-frame coordinates and re-running it keeps the code
// path identical.  Output == ss_encoder_forward on the same fbank up to GEMM summation order
// (tile / split-K choices depend on the row count).
//
// Finality: a frame i of the 40-ms grid reaches, in one layer, keys up to the end of its attention
// chunk and conv taps up to min(i+15, end of its conv chunk); through the subsampler it reaches
// conv1 rows a(i) = min(2i+2, chunk end) and fbank rows b(a(i)).  The final prefix [0, n) is the
// largest one that is closed under "reaches" and whose subsampler inputs all exist.
static int stream_final_rows(int T, int T1, int T2, int k, int achunk, int cchunk, int dwk, int tail) {
  if (achunk <= 0) return 0;                       // full attention: every frame sees the future
  auto reach = [&](int i, int half, int stride) {  // last input row a stride-`stride` conv output i can read
    int r = i * stride + half;
    if (cchunk > 0) r = std::min(r, ((i * stride) / cchunk + 1) * cchunk - 1);
    return r;
  };
  int n = 0;
  for (int i = 0; i < T2; ++i) {                   // subsampler level: frames whose whole cone exists
    const int a = reach(i, k / 2, 2);
    if (a > T1 - 1) break;
    if (reach(a, k / 2, 2) > T - 1 - tail) break;   // the last `tail` fbank frames are not settled yet
    n = i + 1;
  }
  while (n > 0) {                                  // closure under one layer's reach (monotone in i)
    const int i = n - 1;
    const int e_att = (i / achunk + 1) * achunk - 1;
    const int e_conv = reach(i, dwk / 2, 1);
    if (std::max(e_att, e_conv) <= n - 1) break;
    --n;
  }
  return n;
}

extern "C" int ss_encoder_stream_reset(ss_model* m) {
  if (!m) return SS_ERR_ARG;
  m->es_final = 0; m->es_achunk = -1; m->es_cchunk = -1;
  return SS_OK;
}

extern "C" int ss_encoder_stream_set_tail(ss_model* m, int unsettled_fbank_frames) {
  if (!m || unsettled_fbank_frames < 0) return SS_ERR_ARG;
  m->es_tail = unsettled_fbank_frames;
  return SS_OK;
}

extern "C" int ss_encoder_stream_forward(ss_model* m, void* stream, const float* d_fbank, int T, int attn_chunk,
                                         int conv_chunk, float* d_enc_out, int32_t* n_final, int32_t* n_computed) {
  if (!m || T <= 0) return SS_ERR_ARG;
  SkScope sk_scope(m->skws);
  hipStream_t s = (hipStream_t)stream;
  const ss_config& c = m->cfg;
  const int d = c.enc_dim, f = c.enc_ffn, k = c.conv_kernel, Ld = c.enc_layers * d, L = c.enc_layers;
  const int T1 = conv_out_len(T, k, 2), T2 = conv_out_len(T1, k, 2);
  if (T2 <= 0 || T2 > c.max_rel_pos) return SS_ERR_CAPACITY;
  const int cchunk = (conv_chunk > 0 && conv_chunk < 999) ? conv_chunk : 0;
  const int achunk_cfg = (attn_chunk > 0 && attn_chunk < 999999) ? attn_chunk : 0;   // as configured (not clipped by T2)
  const int achunk = (achunk_cfg > 0 && achunk_cfg < T2) ? achunk_cfg : 0;
  if (m->es_achunk != achunk_cfg || m->es_cchunk != cchunk) { m->es_final = 0; m->es_achunk = achunk_cfg; m->es_cchunk = cchunk; }
  if (m->es_final > T2) m->es_final = 0;           // audio got shorter: a new utterance without reset
  if (m->es_cap < T2) {                             // grow (contents are only needed below es_final: keep them)
    const int cap = std::min(c.max_rel_pos, std::max(2 * T2, 256));
    DevBuf nq, ng, no;
    RET(nq.ensure((size_t)L * cap * 3 * d * sizeof(float)));
    RET(ng.ensure((size_t)L * cap * d * sizeof(float)));
    RET(no.ensure((size_t)cap * d * sizeof(float)));
    if (m->es_final > 0) {
      for (int l = 0; l < L; ++l) {
        SS_HIP_CHECK(hipMemcpyAsync(nq.f() + (size_t)l * cap * 3 * d, m->es_qkv.f() + (size_t)l * m->es_cap * 3 * d,
                                    (size_t)m->es_final * 3 * d * sizeof(float), hipMemcpyDeviceToDevice, s));
        SS_HIP_CHECK(hipMemcpyAsync(ng.f() + (size_t)l * cap * d, m->es_glu.f() + (size_t)l * m->es_cap * d,
                                    (size_t)m->es_final * d * sizeof(float), hipMemcpyDeviceToDevice, s));
      }
      SS_HIP_CHECK(hipMemcpyAsync(no.f(), m->es_out.f(), (size_t)m->es_final * d * sizeof(float), hipMemcpyDeviceToDevice, s));
      SS_HIP_CHECK(hipStreamSynchronize(s));
    }
    m->es_qkv.release(); m->es_glu.release(); m->es_out.release();
    m->es_qkv = nq; m->es_glu = ng; m->es_out = no;
    nq.p = nullptr; ng.p = nullptr; no.p = nullptr;
    m->es_cap = cap;
  }
  const int cap = m->es_cap;
  const int r0 = m->es_final;                       // first row to (re)compute
  const int n = T2 - r0;
  if (n_computed) *n_computed = n;

  const size_t n_h1 = (size_t)T1 * (c.conv_channels / 2);
  const size_t n_x = (size_t)T2 * d, n_f = (size_t)n * f;
  RET(m->ws.ensure((n_h1 + 3 * n_x + n_f) * sizeof(float)));
  float* h1 = m->ws.f();
  float* g0 = h1 + n_h1;                // subsampler output [T2, d]
  float* h = g0 + n_x;                  // LN output / attention context (tail rows, indexed from 0)
  float* g2 = h + n_x;                  // depthwise output, absolute rows
  float* ff = g2 + n_x;                 // FFN hidden (tail rows)
  float* x = d_enc_out + (size_t)r0 * d;  // running activations of the tail rows live in the output buffer

  {
    GemmArgs a;
    a.A = d_fbank; a.lda = c.input_feat; a.W = m->sub0.w; a.bias = m->sub0.b; a.C = h1; a.ldc = c.conv_channels / 2;
    a.M = T1; a.N = c.conv_channels; a.Cin = c.input_feat; a.taps = k; a.stride = 2; a.pad = k / 2;
    a.in_len = T; a.chunk = cchunk; a.glu = 1;
    RET(launch_conv_gemm(a, s));
    GemmArgs b;
    b.A = h1; b.lda = c.conv_channels / 2; b.W = m->sub1.w; b.bias = m->sub1.b; b.C = g0; b.ldc = d;
    b.M = T2; b.N = 2 * d; b.Cin = c.conv_channels / 2; b.taps = k; b.stride = 2; b.pad = k / 2;
    b.in_len = T1; b.chunk = cchunk; b.glu = 1;
    RET(launch_conv_gemm(b, s));
  }
  if (r0 > 0)
    SS_HIP_CHECK(hipMemcpyAsync(d_enc_out, m->es_out.f(), (size_t)r0 * d * sizeof(float), hipMemcpyDeviceToDevice, s));
  if (n > 0) {
    RET(linear(s, g0 + (size_t)r0 * d, d, n, m->enc_linear, d, d, x, d));
    const float* P = m->pos_proj.f() + (size_t)(c.max_rel_pos - T2) * Ld;
    for (int l = 0; l < L; ++l) {
      const EncLayer& e = m->enc[l];
      float* qkv = m->es_qkv.f() + (size_t)l * cap * 3 * d;      // absolute rows
      float* glu = m->es_glu.f() + (size_t)l * cap * d;
      RET(ln_linear(s, x, n, e.ffn1_ln, e.ffn1_w1, f, d, ff, f, h, ACT_SILU));
      RET(linear(s, ff, f, n, e.ffn1_w2, d, f, x, d, ACT_NONE, 0.5f, x, d));
      RET(ln_linear(s, x, n, e.attn_ln, e.qkv, 3 * d, d, qkv + (size_t)r0 * 3 * d, 3 * d, h));
      AttnArgs at;
      at.Q = qkv + (size_t)r0 * 3 * d; at.K = qkv + d; at.V = qkv + 2 * d; at.ldq = at.ldk = at.ldv = 3 * d;
      at.O = h; at.ldo = d; at.Tq = n; at.Tk = T2; at.q0 = r0; at.H = c.enc_heads; at.scale = 0.125f;
      at.chunk = achunk; at.P = P + (size_t)l * d; at.ldp = Ld; at.bias_u = e.u; at.bias_v = e.v;
      RET(bind_attn_split(m, at, s));
      RET(launch_attention(at, s));
      RET(linear(s, h, d, n, e.out, d, d, x, d, ACT_NONE, 1.f, x, d));
      RET(ln_linear(s, x, n, e.conv_ln, e.pw1, 2 * d, d, glu + (size_t)r0 * d, d, h, ACT_NONE, 1.f, 1));
      RET(launch_dwconv_bn_silu(glu, d, g2, d, e.dw_wt, c.dw_kernel, e.bn_mean, e.bn_var, e.bn_g, e.bn_b, 1e-5f,
                                T2, d, cchunk, s, nullptr, 0, r0));
      RET(linear(s, g2 + (size_t)r0 * d, d, n, e.pw2, d, d, x, d, ACT_NONE, 1.f, x, d));
      RET(ln_linear(s, x, n, e.ffn2_ln, e.ffn2_w1, f, d, ff, f, h, ACT_SILU));
      RET(linear(s, ff, f, n, e.ffn2_w2, d, f, x, d, ACT_NONE, 0.5f, x, d));
      RET(layernorm(s, x, x, e.final_ln, n, d));
    }
  }
  const int nf = std::max(r0, stream_final_rows(T, T1, T2, k, achunk_cfg, cchunk, c.dw_kernel, m->es_tail));
  if (nf > r0)
    SS_HIP_CHECK(hipMemcpyAsync(m->es_out.f() + (size_t)r0 * d, d_enc_out + (size_t)r0 * d, (size_t)(nf - r0) * d * sizeof(float),
                                hipMemcpyDeviceToDevice, s));
  m->es_final = nf;
  if (n_final) *n_final = nf;
  return SS_OK;
}

// ---- CTC heads ---------------------------------------------------------------------------------
extern "C" int ss_ctc_greedy(ss_model* m, void* stream, int head, const float* d_enc_out, int Tp,
                             int32_t* d_raw, int32_t* d_tokens, int32_t* d_index, int32_t* d_count,
                             float* d_logits) {
  if (!m || Tp <= 0 || head < 0 || head > 1) return SS_ERR_ARG;
  SkScope sk_scope(m->skws);
  hipStream_t s = (hipStream_t)stream;
  const ss_config& c = m->cfg;
  const int V = head == 0 ? c.src_vocab : c.tgt_vocab;
  float* logits = d_logits;
  if (!logits) {
    RET(m->mt_ws.ensure((size_t)Tp * V * sizeof(float)));
    logits = m->mt_ws.f();
  }
  RET(linear(s, d_enc_out, c.enc_dim, Tp, head == 0 ? m->ctc_asr : m->ctc_st, V, c.enc_dim, logits, V));
  RET(launch_masked_argmax(logits, V, Tp, V, c.pad, c.unk, -1, -1, d_raw, s));
  return launch_ctc_collapse(d_raw, Tp, 0, c.pad, d_tokens, d_index, d_count, s);
}

// ---- transformer layers shared by MT decoder / T2U encoder / unit decoder ----------------------
// x [n, D] in place.  self K/V cache rows live in `selfbuf` ([*, 3D], row = absolute position).
// One pre-LN transformer layer on the residual stream x [n, D].  The fused QKV rows are written to
// `qkv_rows` with row stride ld_qkv (straight into a KV cache when decoding); the caller prepares
// the attention descriptors (single utterance or ragged batch) -- their O is `h`, cross Q is `q2`.
static int dec_layer_ex(hipStream_t s, const ss_config& c, const DecLayer& L, float* x, int n, float* qkv_rows,
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
static int dec_layer(hipStream_t s, const ss_config& c, const DecLayer& L, float* x, int n, int pos0,
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

extern "C" int ss_mt_begin(ss_model* m, void* stream, const float* d_enc_out, int Tp) {
  if (!m || Tp <= 0) return SS_ERR_ARG;
  SkScope sk_scope(m->skws);
  hipStream_t s = (hipStream_t)stream;
  const ss_config& c = m->cfg;
  const int D = c.dec_dim;
  RET(m->mt_cross.ensure((size_t)c.mt_layers * Tp * 2 * D * sizeof(float)));
  for (int l = 0; l < c.mt_layers; ++l)
    RET(linear(s, d_enc_out, c.enc_dim, Tp, m->mt[l].cross_kv, 2 * D, c.enc_dim,
               m->mt_cross.f() + (size_t)l * Tp * 2 * D, 2 * D));
  m->mt_Tp = Tp;
  m->mt_len = 0;
  m->mt_enc = d_enc_out;
  return SS_OK;
}

// Bounded-wait time-outs of persistent MT decode steps, process-wide (reported by ss_debug_sk_errors next to the stream-K
// counters).  The step has its OWN device error word per context (it used to alias the stream-K time-out counter,
// so one MT time-out made every later stream-K launch of that workspace bail out); the host collects and clears it whenever
// the persistent form is switched (which both fall-back paths do).
static std::atomic<int> g_mt_timeouts{0};
static unsigned* mt_err_word(ss_model* m) {
  return reinterpret_cast<unsigned*>(static_cast<char*>(m->mt_gran.p) + mt_step_granule_bytes() + 16);
}
static int mt_collect_errors(ss_model* m) {
  if (!m->mt_gran.p) return SS_OK;
  // on the stream the persistent step was last launched on (torch streams are non-blocking: the legacy null stream orders
  // nothing against them -- ADVICE r4): the read sees every earlier launch's error word, the clear lands before any later one
  hipStream_t s = m->mt_last_stream;
  unsigned e = 0;
  SS_HIP_CHECK(hipMemcpyAsync(&e, mt_err_word(m), sizeof(e), hipMemcpyDeviceToHost, s));
  SS_HIP_CHECK(hipStreamSynchronize(s));
  if (e) {
    g_mt_timeouts.fetch_add((int)e, std::memory_order_relaxed);
    SS_HIP_CHECK(hipMemsetAsync(mt_err_word(m), 0, sizeof(unsigned), s));
    SS_HIP_CHECK(hipStreamSynchronize(s));
  }
  return SS_OK;
}

extern "C" int ss_mt_set_persistent(ss_model* m, int workgroups) {
  if (!m || !(workgroups == 0 || workgroups == 64 || workgroups == 128 || workgroups == 256)) return SS_ERR_ARG;
  if (m->mt_persistent > 0) RET(mt_collect_errors(m));
  m->mt_persistent = workgroups;
  return SS_OK;
}

extern "C" int ss_mt_get_persistent(ss_model* m) { return m ? m->mt_persistent : SS_ERR_ARG; }

extern "C" int ss_debug_mt_inject_timeout(ss_model* m) {
  if (!m) return SS_ERR_ARG;
  m->mt_inject_timeout = 1;
  return SS_OK;
}

extern "C" int ss_mt_truncate(ss_model* m, int len) {
  if (!m || len < 0 || len > m->mt_len) return SS_ERR_ARG;
  m->mt_len = len;
  return SS_OK;
}

// Argument block of the persistent decode kernel (mt_step.hip) for context `m`: weights, caches, granules, error word.
static int mt_step_args(ss_model* m, MtStepArgs& a, float* scratch_feats) {
  const ss_config& c = m->cfg;
  const int D = c.dec_dim;
  for (int l = 0; l < MT_L; ++l) {
    const DecLayer& L = m->mt[l];
    if (!L.has_cross) return SS_ERR_ARG;
    MtLayerW& w = a.L[l];
    w.ln1_g = L.self_ln.g; w.ln1_b = L.self_ln.b; w.wqkv = L.self_qkv.w; w.bqkv = L.self_qkv.b; w.wo = L.self_out.w; w.bo = L.self_out.b;
    w.ln2_g = L.cross_ln.g; w.ln2_b = L.cross_ln.b; w.wcq = L.cross_q.w; w.bcq = L.cross_q.b; w.wco = L.cross_out.w; w.bco = L.cross_out.b;
    w.ln3_g = L.ffn_ln.g; w.ln3_b = L.ffn_ln.b; w.w1 = L.fc1.w; w.b1 = L.fc1.b; w.w2 = L.fc2.w; w.b2 = L.fc2.b;
    w.selfbuf = m->mt_self.f() + (size_t)l * c.max_tgt_pos * 3 * D;
    w.cross = m->mt_cross.f() + (size_t)l * m->mt_Tp * 2 * D;
  }
  a.lnf_g = m->mt_ln.g; a.lnf_b = m->mt_ln.b; a.emb = m->mt_emb; a.pos_table = m->mt_pos;
  a.feats = scratch_feats;
  a.gran = reinterpret_cast<mt_u64*>(m->mt_gran.p); a.err = mt_err_word(m);
  a.Tp = m->mt_Tp; a.V = c.tgt_vocab; a.pad = c.pad; a.eos = c.eos;
  a.emb_scale = sqrtf((float)D);
  return SS_OK;
}
// n consecutive epochs for a launch of n steps (0 is what a fresh granule holds: skipped)
static unsigned mt_next_epochs(ss_model* m, int n) {
  if (m->mt_epoch + (unsigned)n + 1u < m->mt_epoch || m->mt_epoch == 0u) m->mt_epoch = 0u;      // wrap: start over above 0
  const unsigned first = m->mt_epoch + 1u;
  m->mt_epoch += (unsigned)n;
  return first;
}
static int mt_inject(ss_model* m, MtStepArgs& a, hipStream_t s) {      // test hook: this one launch sees a time-out that already happened
  if (!m->mt_inject_timeout) return SS_OK;
  m->mt_inject_timeout = 0;
  a.err = reinterpret_cast<unsigned*>(static_cast<char*>(m->mt_gran.p) + mt_step_granule_bytes());
  SS_HIP_CHECK(hipMemsetAsync(a.err, 0x01, sizeof(unsigned), s));
  return SS_OK;
}
static bool mt_persistent_ok(const ss_model* m) {
  const ss_config& c = m->cfg;
  return m->mt_persistent > 0 && c.mt_layers == MT_L && c.dec_dim == MT_D && c.dec_ffn == MT_F && c.dec_heads == MT_H;
}
static int mt_gran_ensure(ss_model* m, hipStream_t s) {
  if (!m->mt_gran.p) {
    RET(m->mt_gran.ensure(mt_step_granule_bytes() + 64));   // + one spare error word (ss_debug_mt_inject_timeout) + the step's own
    SS_HIP_CHECK(hipMemsetAsync(m->mt_gran.p, 0, mt_step_granule_bytes() + 64, s));
    SS_HIP_CHECK(hipStreamSynchronize(s));
  }
  return SS_OK;
}

extern "C" int ss_mt_append(ss_model* m, void* stream, const int32_t* d_tokens, int n, int pos0, int ban_eos,
                            int force_eos, float* d_feats, int32_t* d_next, int n_tail_pad) {
  if (!m || n <= 0 || pos0 < 0 || pos0 > m->mt_len || m->mt_Tp <= 0) return SS_ERR_ARG;
  SkScope sk_scope(m->skws);
  const ss_config& c = m->cfg;
  if (pos0 + n + 2 > c.max_tgt_pos) return SS_ERR_CAPACITY;
  hipStream_t s = (hipStream_t)stream;
  const int D = c.dec_dim, F = c.dec_ffn, V = c.tgt_vocab;
  RET(m->mt_ws.ensure(((size_t)n * (4 * D + F) + V) * sizeof(float)));
  float* x = m->mt_ws.f();
  float* h = x + (size_t)n * D;
  float* q2 = h + (size_t)n * D;
  float* feats = q2 + (size_t)n * D;
  float* ff = feats + (size_t)n * D;
  float* logits = ff + (size_t)n * F;
  if (mt_persistent_ok(m) && n == 1 && d_next && n_tail_pad == 0) {
    // one persistent launch for the whole step (mt_step.hip; opt-in)
    RET(mt_gran_ensure(m, s));
    MtStepArgs a;
    RET(mt_step_args(m, a, feats));
    a.tok = d_tokens; a.feats = d_feats ? d_feats : feats; a.next = d_next;
    a.pos0 = pos0; a.n_steps = 1;
    a.min_len = ban_eos ? pos0 + 1 : 0;                   // </s> banned at positions < min_len, forced at positions >= max_len
    a.max_len = force_eos ? pos0 : 0x7fffffff;
    RET(mt_inject(m, a, s));
    a.epoch = mt_next_epochs(m, 1);
    m->mt_last_stream = s;
    RET(launch_mt_step(a, m->mt_persistent, s));
    m->mt_len = pos0 + n;
    return SS_OK;
  }
  // sqrt(D) * E[tok] + sinusoid(position), positions start at padding_idx + 1 (transformer_decoder.py:297-326)
  RET(launch_embed_tokens(d_tokens, m->mt_emb, m->mt_pos, sqrtf((float)D), pos0 + c.pad + 1, x, n, D, s, 1, c.pad, V));
  for (int l = 0; l < c.mt_layers; ++l) {
    float* selfbuf = m->mt_self.f() + (size_t)l * c.max_tgt_pos * 3 * D;
    const float* cross = m->mt_cross.f() + (size_t)l * m->mt_Tp * 2 * D;
    RET(dec_layer(s, c, m->mt[l], x, n, pos0, selfbuf, true, cross, m->mt_Tp, h, q2, ff, n_tail_pad, 0));
  }
  float* fo = d_feats ? d_feats : feats;
  m->mt_len = pos0 + n;
  Lin proj{m->mt_emb, nullptr};  // tied output projection, no bias
  GemmArgs g;                    // decode step (one new token): final LayerNorm in the prologue of the vocabulary GEMV,
  g.A = x; g.lda = D; g.W = proj.w; g.C = logits; g.ldc = V; g.M = 1; g.N = V; g.Cin = D; g.in_len = 1; g.same_rows = 1;
  g.ln_g = m->mt_ln.g; g.ln_b = m->mt_ln.b; g.ln_out = fo;   // which also writes the features row: one launch fewer per step
  if (n == 1 && d_next && gemv_eligible(g) && D <= 512 && !g_no_mt_ln_fusion) {
    RET(launch_conv_gemm(g, s));
    return launch_masked_argmax(logits, V, 1, V, c.pad, ban_eos ? c.eos : -1, -1, force_eos ? c.eos : -1, d_next, s);
  }
  RET(launch_layernorm(x, D, fo, D, m->mt_ln.g, m->mt_ln.b, n, D, 1e-5f, s));
  if (d_next) {
    RET(linear(s, fo + (size_t)(n - 1) * D, D, 1, proj, V, D, logits, V));
    RET(launch_masked_argmax(logits, V, 1, V, c.pad, ban_eos ? c.eos : -1, -1, force_eos ? c.eos : -1, d_next, s));
  }
  return SS_OK;
}


// Whole beam-1 search in one call: the token chain lives on the device (each step's argmax writes
// the next step's input), the host only peeks at it every kCheck steps to notice </s>.
extern "C" int ss_mt_greedy(ss_model* m, void* stream, const float* d_enc_out, int Tp, const int32_t* h_prefix,
                            int n_prefix, int max_len, int min_len, int32_t* h_out_tokens, int* h_n_out,
                            float* d_feats, int* h_n_feats) {
  if (!m || !d_enc_out || Tp <= 0 || n_prefix < 0 || max_len < n_prefix || !h_out_tokens || !h_n_out || !d_feats)
    return SS_ERR_ARG;
  SkScope sk_scope(m->skws);
  const ss_config& c = m->cfg;
  if (max_len + 3 > c.max_tgt_pos) return SS_ERR_CAPACITY;
  for (int i = 0; i < n_prefix; ++i)
    if (h_prefix[i] < 0 || h_prefix[i] >= c.tgt_vocab) return SS_ERR_ARG;   // nn.Embedding's IndexError in the reference
  hipStream_t s = (hipStream_t)stream;
  constexpr int kCheck = 4;
  const int D = c.dec_dim;
  RET(ss_mt_begin(m, stream, d_enc_out, Tp));
  int32_t* tok = reinterpret_cast<int32_t*>(m->mt_tok.p);
  int32_t* host = m->mt_tok_host;
  host[0] = c.eos;
  for (int i = 0; i < n_prefix; ++i) host[1 + i] = h_prefix[i];
  const int start = n_prefix;
  SS_HIP_CHECK(hipMemcpyAsync(tok, host, (size_t)(start + 1) * sizeof(int32_t), hipMemcpyHostToDevice, s));
  if (mt_persistent_ok(m) && !g_no_mt_device_loop) {
    // ---- the whole search as ONE persistent launch (device-side token loop, mt_step.hip): positions `first` .. max_len, the loop
    // ends at </s>; with a prefix its tokens are fed first in one launch-per-op pass ----
    int first = 0;
    if (start > 0) {
      RET(ss_mt_append(m, stream, tok, start + 1, 0, start < min_len, start >= max_len, d_feats, tok + start + 1, 0));
      first = start + 1;
    }
    const int n_steps = max_len + 1 - first;
    if (n_steps > 0) {
      RET(mt_gran_ensure(m, s));
      MtStepArgs a;
      RET(mt_step_args(m, a, d_feats));
      a.tok = tok + first; a.feats = d_feats + (size_t)first * D; a.next = tok + first + 1;
      a.pos0 = first; a.n_steps = n_steps; a.min_len = min_len; a.max_len = max_len;
      RET(mt_inject(m, a, s));
      a.epoch = mt_next_epochs(m, n_steps);
      m->mt_last_stream = s;
      RET(launch_mt_step(a, m->mt_persistent, s));
    }
    const int lo = start + 1, hi = max_len + 1;             // generated tokens live at chain indices lo .. hi
    SS_HIP_CHECK(hipMemcpyAsync(host + lo, tok + lo, (size_t)(hi - lo + 1) * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    SS_HIP_CHECK(hipStreamSynchronize(s));
    int eos_at = -1;
    for (int i = lo; i <= hi && eos_at < 0; ++i) {
      if (host[i] < 0) {                                    // a bounded wait of the persistent kernel timed out
        fprintf(stderr, "streamspeech_hip: persistent MT decode step timed out (its %d workgroups were not all resident); "
                        "this context falls back to one launch per op\n", m->mt_persistent);
        RET(mt_collect_errors(m));
        m->mt_persistent = 0;
        return ss_mt_greedy(m, stream, d_enc_out, Tp, h_prefix, n_prefix, max_len, min_len, h_out_tokens, h_n_out, d_feats, h_n_feats);
      }
      if (host[i] == c.eos) eos_at = i;
    }
    const int end = eos_at >= 0 ? eos_at : hi;              // (position max_len forces </s>: eos_at is always found)
    const int n_out = end - start;
    for (int i = 0; i < n_out; ++i) h_out_tokens[i] = host[start + 1 + i];
    *h_n_out = n_out;
    if (h_n_feats) *h_n_feats = end;
    m->mt_len = end;
    return SS_OK;
  }
  // step `start`: feed [eos, prefix...] in one pass
  RET(ss_mt_append(m, stream, tok, start + 1, 0, start < min_len, start >= max_len, d_feats, tok + start + 1, 0));
  int step = start + 1;      // next position to feed == index of the newest generated token
  int eos_at = -1, checked = start + 1;
  while (true) {
    const bool last = step > max_len;
    if (last || (step - (start + 1)) % kCheck == kCheck - 1) {
      SS_HIP_CHECK(hipMemcpyAsync(host + checked, tok + checked, (size_t)(step + 1 - checked) * sizeof(int32_t),
                                  hipMemcpyDeviceToHost, s));
      SS_HIP_CHECK(hipStreamSynchronize(s));
      for (int i = checked; i <= step && eos_at < 0; ++i) {
        if (host[i] < 0 && m->mt_persistent > 0) {          // mt_step.hip: a bounded wait of the persistent step timed out
          fprintf(stderr, "streamspeech_hip: persistent MT decode step timed out (its %d workgroups were not all resident); "
                          "this context falls back to one launch per op\n", m->mt_persistent);
          RET(mt_collect_errors(m));
          m->mt_persistent = 0;
          return ss_mt_greedy(m, stream, d_enc_out, Tp, h_prefix, n_prefix, max_len, min_len, h_out_tokens, h_n_out, d_feats,
                              h_n_feats);
        }
        if (host[i] == c.eos) eos_at = i;
      }
      checked = step + 1;
      if (eos_at >= 0 || last) break;
    }
    RET(ss_mt_append(m, stream, tok + step, 1, step, step < min_len, step >= max_len,
                     d_feats + (size_t)step * D, tok + step + 1, 0));
    ++step;
  }
  const int end = eos_at >= 0 ? eos_at : step;          // index of the last generated token
  const int n_out = end - start;                          // tokens after the prefix (incl. a final eos)
  for (int i = 0; i < n_out; ++i) h_out_tokens[i] = host[start + 1 + i];
  *h_n_out = n_out;
  if (h_n_feats) *h_n_feats = end;                        // fed positions 0 .. end-1 hold valid features
  m->mt_len = end;
  return SS_OK;
}

// ---- T2U encoder + CTC unit decoder ------------------------------------------------------------
extern "C" int ss_t2u_units(ss_model* m, void* stream, const float* d_mt_feats, int n, int t2u_causal,
                            int mask_eos, int32_t* d_raw, int32_t* d_tokens, int32_t* d_count, float* d_logits,
                            int n_tail_pad) {
  if (!m || n <= 0) return SS_ERR_ARG;
  SkScope sk_scope(m->skws);
  hipStream_t s = (hipStream_t)stream;
  const ss_config& c = m->cfg;
  const int D = c.dec_dim, F = c.dec_ffn, U = n * c.ctc_upsample, V = c.unit_vocab;
  const size_t nx = (size_t)U * D;
  // t2u: x,h,self(3D) on n rows; unit: x,h,q2 on U rows, self 3D on U rows, ff U*F, cross kv n*2D, logits U*V
  const size_t total = 3 * nx + (size_t)U * 3 * D + (size_t)U * F + (size_t)n * 2 * D + (size_t)n * D +
                       (d_logits ? 0 : (size_t)U * V) + (size_t)U;
  RET(m->ws.ensure(total * sizeof(float)));
  float* x = m->ws.f();
  float* h = x + nx;
  float* q2 = h + nx;
  float* selfbuf = q2 + nx;
  float* ff = selfbuf + (size_t)U * 3 * D;
  float* crosskv = ff + (size_t)U * F;
  float* t2u_out = crosskv + (size_t)n * 2 * D;
  float* logits = d_logits ? d_logits : t2u_out + (size_t)n * D;
  int32_t* idx_scratch = reinterpret_cast<int32_t*>((d_logits ? t2u_out + (size_t)n * D : logits + (size_t)U * V));

  // T2U encoder (transformer_encoder.py:32-77): 2 pre-LN layers + final LN
  SS_HIP_CHECK(hipMemcpyAsync(x, d_mt_feats, (size_t)n * D * sizeof(float), hipMemcpyDeviceToDevice, s));
  for (int l = 0; l < c.t2u_layers; ++l)
    RET(dec_layer(s, c, m->t2u[l], x, n, 0, selfbuf, t2u_causal != 0, nullptr, 0, h, q2, ff, n_tail_pad, 0));
  RET(launch_layernorm(x, D, t2u_out, D, m->t2u_ln.g, m->t2u_ln.b, n, D, 1e-5f, s));
  // unit decoder input: each T2U state 25x + the (quirky) positional row
  RET(launch_upsample_add_pos(t2u_out, n, c.ctc_upsample, m->unit_pos_row, (float)c.pad, x, D, s));
  for (int l = 0; l < c.unit_layers; ++l) {
    RET(linear(s, t2u_out, D, n, m->unit[l].cross_kv, 2 * D, D, crosskv, 2 * D));
    RET(dec_layer(s, c, m->unit[l], x, U, 0, selfbuf, true, crosskv, n, h, q2, ff, n_tail_pad * c.ctc_upsample, n_tail_pad));
  }
  RET(launch_layernorm(x, D, h, D, m->unit_ln.g, m->unit_ln.b, U, D, 1e-5f, s));
  RET(linear(s, h, D, U, m->unit_out, V, D, logits, V));
  RET(launch_masked_argmax(logits, V, U, V, c.pad, c.unk, mask_eos ? c.eos : -1, -1, d_raw, s));
  return launch_ctc_collapse(d_raw, U, V - 1, c.pad, d_tokens, idx_scratch, d_count, s);
}

// =================================================================================================
// vocoder
// =================================================================================================
struct ConvW { const float* w = nullptr; const float* b = nullptr; const float* ww = nullptr; };   // ww: Winograd form (64-channel stage ResBlock convs)
struct ss_vocoder {
  ss_vocoder_config cfg;
  WeightTable wt;
  SkWorkspace* skws = nullptr;       // stream-K hand-off state of this context (freed with the handle)
  const float* dict = nullptr;
  ConvW dur_c1, dur_c2, dur_proj, pre, post;
  LN dur_ln1, dur_ln2;
  std::vector<ConvW> ups;
  std::vector<ConvW> rb_c1, rb_c2;  // [(stage*n_res + j)*3 + d]
  float* wino = nullptr;             // Winograd F(2,3) forms of the 32- / 64- / 128-channel stages' ResBlock conv weights (conv_c64w.hip):
  const float* wino_key = nullptr;   // ONE buffer per weight blob, shared by every context over that blob (wino_share below)
  DevBuf ws, small, segs;
  int x3 = 0;          // split-bf16 contraction of the C >= 64 generator convs (ss_vocoder_set_bf16x3); default off = exact f32
};

// Transformed weights are a function of the weight blob alone: contexts made over the same blob (HipVocoder.new_context: one per
// concurrent stream) borrow one buffer instead of packing ~16 MB each.  Keyed by (device, blob pointer), ref-counted.
namespace {
struct WinoShared { DevBuf buf; size_t floats = 0; int refs = 0; };
std::mutex g_wino_mu;
std::map<std::pair<int, const float*>, WinoShared> g_wino;
}  // namespace

extern "C" int ss_vocoder_create(const ss_vocoder_config* cfg, const float* d_blob, size_t blob_floats,
                                 const char* const* names, const int64_t* offsets, const int64_t* numels,
                                 int n_slots, ss_vocoder** out) {
  if (!cfg || !d_blob || !out || cfg->n_up > 8 || cfg->n_res > 4) return SS_ERR_ARG;
  ss_vocoder* v = new ss_vocoder();
  v->cfg = *cfg;
  v->skws = sk_workspace_new();
  int rc = v->wt.build(d_blob, blob_floats, names, offsets, numels, n_slots);
  if (rc != SS_OK) { sk_workspace_free(v->skws); delete v; return rc; }
  WeightTable& w = v->wt;
  const int E = cfg->embedding_dim, Hd = cfg->dur_hidden, kd = cfg->dur_kernel;
  v->dict = w.get("voc.dict", (int64_t)cfg->num_embeddings * E);
  v->dur_c1 = {w.get("voc.dur.conv1.w", (int64_t)Hd * kd * E), w.get("voc.dur.conv1.b", Hd)};
  v->dur_ln1 = {w.get("voc.dur.ln1.g", Hd), w.get("voc.dur.ln1.b", Hd)};
  v->dur_c2 = {w.get("voc.dur.conv2.w", (int64_t)Hd * kd * Hd), w.get("voc.dur.conv2.b", Hd)};
  v->dur_ln2 = {w.get("voc.dur.ln2.g", Hd), w.get("voc.dur.ln2.b", Hd)};
  v->dur_proj = {w.get("voc.dur.proj.w", Hd), w.get("voc.dur.proj.b", 1)};
  const int C0 = cfg->upsample_initial_channel;
  v->pre = {w.get("voc.pre.w", (int64_t)C0 * 7 * cfg->model_in_dim), w.get("voc.pre.b", C0)};
  int C = C0;
  for (int i = 0; i < cfg->n_up; ++i) {
    const int Co = C / 2, st = cfg->upsample_rates[i];
    v->ups.push_back({w.get("voc.up" + std::to_string(i) + ".w", (int64_t)st * Co * 3 * C),
                      w.get("voc.up" + std::to_string(i) + ".b", (int64_t)st * Co)});
    for (int j = 0; j < cfg->n_res; ++j) {
      const int kr = cfg->resblock_kernel_sizes[j];
      for (int dd = 0; dd < 3; ++dd) {
        const std::string p = "voc.rb" + std::to_string(i * cfg->n_res + j);
        v->rb_c1.push_back({w.get(p + ".c1." + std::to_string(dd) + ".w", (int64_t)Co * kr * Co),
                            w.get(p + ".c1." + std::to_string(dd) + ".b", Co)});
        v->rb_c2.push_back({w.get(p + ".c2." + std::to_string(dd) + ".w", (int64_t)Co * kr * Co),
                            w.get(p + ".c2." + std::to_string(dd) + ".b", Co)});
      }
    }
    C = Co;
  }
  v->post = {w.get("voc.post.w", (int64_t)7 * C), w.get("voc.post.b", 1)};
  if (!w.missing.empty()) { sk_workspace_free(v->skws); delete v; return SS_ERR_MISSING_WEIGHT; }
  {
    // Winograd forms of the 32-, 64- and 128-channel stages' ResBlock convs (conv_c64w.hip), made once per context from the packed weights
    auto wino_stage = [](int ch) { return ch == 32 || ch == 64 || ch == 128 || ch == 256; };
    size_t need = 0;
    int Cs = C0;
    for (int i = 0; i < cfg->n_up; ++i) {
      Cs /= 2;
      if (wino_stage(Cs)) for (int j = 0; j < cfg->n_res; ++j) need += 6 * (size_t)Cs * ((cfg->resblock_kernel_sizes[j] + 2) / 3) * 4 * Cs;
    }
    if (need) {
      int dev = 0;
      if (hipGetDevice(&dev) != hipSuccess) { sk_workspace_free(v->skws); delete v; return SS_ERR_HIP; }
      std::lock_guard<std::mutex> lk(g_wino_mu);          // (held over the pack: a second context of the same blob waits for it)
      WinoShared& sh = g_wino[std::make_pair(dev, d_blob)];
      const bool fresh = sh.refs == 0 || sh.floats != need;
      if (fresh && sh.refs > 0) { g_wino.erase(std::make_pair(dev, d_blob)); sk_workspace_free(v->skws); delete v; return SS_ERR_ARG; }   // same blob, another config
      if (fresh) {
        rc = sh.buf.ensure(need * sizeof(float));
        if (rc != SS_OK) { g_wino.erase(std::make_pair(dev, d_blob)); sk_workspace_free(v->skws); delete v; return rc; }
        sh.floats = need;
      }
      float* dst = sh.buf.f();
      Cs = C0;
      for (int i = 0; i < cfg->n_up && rc == SS_OK; ++i) {
        Cs /= 2;
        if (!wino_stage(Cs)) continue;
        for (int j = 0; j < cfg->n_res && rc == SS_OK; ++j) {
          const int kr = cfg->resblock_kernel_sizes[j];
          const size_t n = (size_t)Cs * ((kr + 2) / 3) * 4 * Cs;
          for (int dd = 0; dd < 3 && rc == SS_OK; ++dd) {
            const int idx = (i * cfg->n_res + j) * 3 + dd;
            if (fresh) rc = launch_wino_pack(v->rb_c1[idx].w, dst, Cs, kr, nullptr);
            v->rb_c1[idx].ww = dst; dst += n;
            if (fresh && rc == SS_OK) rc = launch_wino_pack(v->rb_c2[idx].w, dst, Cs, kr, nullptr);
            v->rb_c2[idx].ww = dst; dst += n;
          }
        }
      }
      // the pack ran on the null stream: d_blob must be device-visible when ss_vocoder_create is called (header), and the buffer is
      // complete for every stream once this returns
      if (fresh && rc == SS_OK && hipDeviceSynchronize() != hipSuccess) rc = SS_ERR_HIP;
      if (rc != SS_OK) {
        if (fresh) { sh.buf.release(); g_wino.erase(std::make_pair(dev, d_blob)); }
        sk_workspace_free(v->skws); delete v; return rc;
      }
      ++sh.refs;
      v->wino = sh.buf.f(); v->wino_key = d_blob;
    }
  }
  *out = v;
  return SS_OK;
}

extern "C" int ss_vocoder_set_bf16x3(ss_vocoder* v, int on) {
  if (!v) return SS_ERR_ARG;
  v->x3 = on ? 1 : 0;
  return SS_OK;
}

extern "C" void ss_vocoder_destroy(ss_vocoder* v) {
  if (!v) return;
  v->ws.release(); v->small.release(); v->segs.release();
  if (v->wino_key) {
    std::lock_guard<std::mutex> lk(g_wino_mu);
    for (auto it = g_wino.begin(); it != g_wino.end(); ++it)
      if (it->first.second == v->wino_key && it->second.buf.f() == v->wino) {
        if (--it->second.refs == 0) { it->second.buf.release(); g_wino.erase(it); }
        break;
      }
  }
  sk_workspace_free(v->skws);
  delete v;
}

// -------------------------------------------------------------------------------------------------
// HiFi-GAN generator stack shared by the single-utterance and the ragged-batch entry points
// (hifigan.py:154-170).  `conv(GemmArgs&, scale)` fills in the row geometry (rows = frames * scale,
// segment table) and launches; `on_stage(scale)` is called when the row scale changes.
//
// Leaky-ReLU placement: the reference applies leaky_relu to the *input* of every conv.  On the
// MFMA-bound stages (C >= 64) the producer writes the activated tensor instead (conv1: ACT_LRELU
// epilogue; conv2 / up-conv: second output C2 = leaky_relu(C)), so the consumer's MFMA loop carries
// no VALU work; the values are bit-identical (same f32 select on the same f32 number).  On the
// HBM-bound late stages (C < 64) the extra write would cost more than the VALU, so the activation
// stays on the consumer's A-fragment path there.
// -------------------------------------------------------------------------------------------------
struct GenBufs { float *bx, *bt, *br, *bs, *bxa, *bra, *bsa, *br2; };

template <class ConvFn, class StageFn, class GeomFn>
static int hifigan_stack(const ss_vocoder* v, hipStream_t s, ConvFn&& conv, StageFn&& on_stage, GeomFn&& geom,
                         const float* frames, int Ft, const GenBufs& b, int* out_scale, int* out_C) {
  const ss_vocoder_config& c = v->cfg;
  // Stages with >= 64 channels get their input leaky-ReLU from the PRODUCER (a second, pre-activated output: VALU work inside a
  // stream-K MFMA loop costs matrix-core time) -- except the 64-channel stage of a packed batch, whose convs run on conv_c64.hip:
  // that kernel applies the activation once per element while it stages its input slab, so no twin tensor is written or read.
  // The 32-channel stage of a packed batch runs its convs one by one on conv_c32.hip instead of one fused launch per ResBlock.
  auto slab_stage = [&](int channels) {
    if (v->x3 || !(channels == 64 ? conv_c64_enabled() : channels == 32 ? conv_c32_enabled() : channels == 16 ? conv_c16_enabled() : false)) return false;
    long long rows = Ft; int ch = c.upsample_initial_channel; bool found = false;
    for (int i = 0; i < c.n_up && !found; ++i) { rows *= c.upsample_rates[i]; ch /= 2; found = ch == channels; }
    if (!found || rows >= (1ll << 30)) return false;
    GemmArgs probe;
    probe.same_rows = 1; probe.Cin = probe.N = probe.lda = probe.ldc = channels; probe.taps = 3; probe.dil = 1; probe.pad = 1;
    probe.M = probe.in_len = (int)rows; probe.in_act = ACT_LRELU;
    return channels == 64 ? conv_c64_eligible(probe) : channels == 32 ? conv_c32_eligible(probe) : conv_c16_eligible(probe);
  };
  const bool c64 = slab_stage(64), c32 = slab_stage(32), c16 = slab_stage(16);
  // The 128-channel stage of a packed batch: its ResBlock convs in Winograd form on the slab kernel (conv_c64w.hip at 128 channels), which
  // activates while staging -- so the convs of that stage neither read nor write twins; only the up-conv that LEAVES the stage (on conv_sk2)
  // still reads one, written by the stage's last conv.  Taken only if every conv of the stage is eligible (there is no direct slab form).
  auto wino_slab_stage = [&](int channels) {
    if (v->x3 || !(channels == 128 ? conv_c128w_enabled() : conv_c256w_enabled())) return false;
    long long rows = Ft; int ch = c.upsample_initial_channel, stage = -1;
    for (int i = 0; i < c.n_up && stage < 0; ++i) { rows *= c.upsample_rates[i]; ch /= 2; if (ch == channels) stage = i; }
    if (stage < 0 || rows >= (1ll << 30)) return false;
    int sc = 1, gM = 0, gnseg = 0; const int* gsegs = nullptr;
    for (int i = 0; i <= stage; ++i) sc *= c.upsample_rates[i];
    geom(sc, gM, gsegs, gnseg);                              // the row geometry the stage's launches will carry
    for (int j = 0; j < c.n_res; ++j)
      for (int dd = 0; dd < 3; ++dd)
        for (int which = 0; which < 2; ++which) {
          const int idx = (stage * c.n_res + j) * 3 + dd;
          GemmArgs probe;
          probe.same_rows = 1; probe.Cin = probe.N = probe.lda = probe.ldc = probe.ldr = probe.ldr2 = probe.ldc2 = channels;
          probe.taps = c.resblock_kernel_sizes[j]; probe.dil = which ? 1 : c.resblock_dilations[j][dd];
          probe.pad = probe.dil * (probe.taps - 1) / 2; probe.M = probe.in_len = gM; probe.nseg = gnseg; probe.in_act = ACT_LRELU;
          probe.Wwino = which ? v->rb_c2[idx].ww : v->rb_c1[idx].ww;
          if (!(channels == 128 ? conv_c128w_eligible(probe) : conv_c256w_eligible(probe))) return false;
        }
    return true;
  };
  // (round 5: the 256-channel stage the same way -- conv_c64w.hip at CH = 256: two slab phases of 128 input channels, two column halves)
  const bool c128 = wino_slab_stage(128), c256 = wino_slab_stage(256);
  // does a ResBlock conv of this stage read a pre-activated twin?  (does the producer have to write one?)
  auto preact = [c64, c128, c256](int channels) {
    return channels >= 64 && !(c64 && channels == 64) && !(c128 && channels == 128) && !(c256 && channels == 256);
  };
  // the up-conv that leaves a stage runs on conv_sk2 for >= 128 channels (N = stride x C / 2) and on conv_c64 for the 64-channel stage
  auto up_preact = [c64](int channels) { return channels >= 64 && !(c64 && channels == 64); };
  auto mk = [v](const float* A, int Cin, const ConvW& cw, int Cout, int k, int dil, float* Cc, int ldc) {
    GemmArgs a;
    a.A = A; a.lda = Cin; a.W = cw.w; a.Wwino = cw.ww; a.bias = cw.b; a.C = Cc; a.ldc = ldc; a.ldr = ldc; a.ldr2 = ldc; a.ldc2 = ldc;
    a.N = Cout; a.Cin = Cin; a.taps = k; a.dil = dil; a.stride = 1; a.pad = dil * (k - 1) / 2; a.same_rows = 1;
    a.x3 = v->x3;
    return a;
  };
  int scale = 1, C = c.upsample_initial_channel;
  RET(on_stage(scale));
  {
    GemmArgs a = mk(frames, c.model_in_dim, v->pre, C, 7, 1, b.bx, C);
    if (up_preact(C)) a.C2 = b.bxa;
    RET(conv(a, scale));
  }
  for (int i = 0; i < c.n_up; ++i) {
    const int st = c.upsample_rates[i], Co = C / 2;
    const bool pa_in = up_preact(C), pa = preact(Co);
    {
      // leaky_relu(0.1) -> ConvTranspose1d as a 3-tap polyphase conv with N = st*Co: row q of the
      // [T, st*Co] result is rows q*st .. q*st+st-1 of the [T*st, Co] signal.
      GemmArgs a = mk(pa_in ? b.bxa : b.bx, C, v->ups[i], st * Co, 3, 1, b.bs, st * Co);
      if (!pa_in) { a.in_act = ACT_LRELU; a.in_slope = 0.1f; }
      if (pa) a.C2 = b.bsa;
      a.algo_flops = 2.0 * Ft * scale * C * Co * c.upsample_kernel_sizes[i];   // zero-padded polyphase slots are not work
      RET(conv(a, scale));
    }
    scale *= st; C = Co;
    RET(on_stage(scale));
    const bool pa_next = (i + 1 < c.n_up) && up_preact(C);  // the next up-conv reads leaky_relu(x)
    for (int j = 0; j < c.n_res; ++j) {
      const int kr = c.resblock_kernel_sizes[j];
      // narrow stages: each (dilated conv, plain conv, residual) pair as ONE launch with the intermediate in LDS
      int gM = 0, gnseg = 0; const int* gsegs = nullptr;
      geom(scale, gM, gsegs, gnseg);
      // measured per kernel size (rocprofv3, batch 32): fused wins 20-25 % at k = 3 (HBM-bound), ties at k = 7, loses
      // 10-30 % at k = 11 (MFMA-bound: halo rows of conv1 are extra work and the 54-KB footprint halves the occupancy)
      // narrow stages: the whole ResBlock (three pairs) as ONE persistent launch (resblock.hip); bit-identical to the
      // pair / two-launch forms below, which stay as the A/B and fallback path
      // conv_c32.hip: at k = 11 (MFMA-bound) six separate convs beat the fused ResBlock launch -- no halo recompute: 107 vs 86
      // TFLOP/s in the pipeline; at k = 3 / 7 the fused launch wins (92-98 vs 56-93: the separate convs are HBM-bound there)
      // (round 5: from k = 7 -- a 7-tap conv is 10 instead of 12 MFMA k-blocks per pair since the one-tap tail group: 6243 vs 6200 x RT,
      //  k >= 3: 6221; tools/jobs/r05_m.sh)
      static const int c32_min_k = getenv("SS_CONV_C32_MIN_K") ? atoi(getenv("SS_CONV_C32_MIN_K")) : 7;
      // conv_c16.hip: the same split at 16 channels (weight matrix in registers): +0.5 %; the round-1 slab kernel (weights in LDS) conv by
      // conv measures -0.3 % against the fused launch, profiles/r04_c16_bench.txt + tools/jobs/r04_o.sh / r04_p.sh
      static const int c16_min_k = getenv("SS_CONV_C16_MIN_K") ? atoi(getenv("SS_CONV_C16_MIN_K")) : 11;
      const bool per_conv = (c32 && C == 32 && kr >= c32_min_k) || (c16 && C == 16 && kr >= c16_min_k);
      if (!pa && !per_conv && !disp().no_resblock_fusion && resblock_fused_eligible(C, kr, c.resblock_dilations[j], C, C, gnseg, gM)) {
        const float *W1[3], *B1[3], *W2[3], *B2[3];
        for (int dd = 0; dd < 3; ++dd) {
          const int idx = (i * c.n_res + j) * 3 + dd;
          W1[dd] = v->rb_c1[idx].w; B1[dd] = v->rb_c1[idx].b; W2[dd] = v->rb_c2[idx].w; B2[dd] = v->rb_c2[idx].b;
        }
        RET(launch_resblock_fused(b.bs, C, W1, B1, W2, B2, c.resblock_dilations[j], b.bx, C, j > 0 ? b.bx : nullptr, C,
                                  j == c.n_res - 1 ? (float)c.n_res : 0.f, C, kr, gM, 0.1f, gsegs, gnseg, s));
        continue;
      }
      const bool fuse = !pa && !per_conv && !disp().no_pair_fusion && kr == 3 &&
                        conv_pair_eligible(C, kr, c.resblock_dilations[j][2], C, C, gnseg, gM);
      const float* cur = b.bs;
      for (int dd = 0; dd < 3; ++dd) {
        const int idx = (i * c.n_res + j) * 3 + dd;
        if (fuse) {
          float* out = dd == 0 ? b.br : dd == 1 ? b.br2 : b.bx;
          const float* R2 = (dd == 2 && j > 0) ? b.bx : nullptr;
          const float div = (dd == 2 && j == c.n_res - 1) ? (float)c.n_res : 0.f;
          RET(launch_conv_pair(cur, C, v->rb_c1[idx].w, v->rb_c1[idx].b, v->rb_c2[idx].w, v->rb_c2[idx].b, out, C, R2, C, div,
                               nullptr, C, 0.1f, C, kr, c.resblock_dilations[j][dd], gM, gM, 0.1f, gsegs, gnseg, s));
          cur = out;
          continue;
        }
        const float* rin = dd == 0 ? b.bs : b.br;           // residual stream (un-activated)
        const float* rin_act = dd == 0 ? b.bsa : b.bra;     // its leaky_relu, when pre-activated
        GemmArgs a1 = mk(pa ? rin_act : rin, C, v->rb_c1[idx], C, kr, c.resblock_dilations[j][dd], b.bt, C);
        if (pa) { a1.act = ACT_LRELU; a1.act_slope = 0.1f; }
        else { a1.in_act = ACT_LRELU; a1.in_slope = 0.1f; }
        RET(conv(a1, scale));
        GemmArgs a2 = mk(b.bt, C, v->rb_c2[idx], C, kr, 1, dd < 2 ? b.br : b.bx, C);
        if (!pa) { a2.in_act = ACT_LRELU; a2.in_slope = 0.1f; }
        a2.R = rin;
        if (dd < 2) {
          if (pa) a2.C2 = b.bra;
        } else {
          // last conv of the resblock also folds the MRF sum: xs (+)= resblock_j(x); x = xs / n_res
          a2.R2 = j == 0 ? nullptr : b.bx;
          a2.div = (j == c.n_res - 1) ? (float)c.n_res : 0.f;
          if (pa_next && j == c.n_res - 1) a2.C2 = b.bxa;
        }
        RET(conv(a2, scale));
      }
    }
  }
  *out_scale = scale; *out_C = C;
  return SS_OK;
}

static int conv1d(hipStream_t s, const float* A, int T, int Cin, const ConvW& cw, int Cout, int k, int dil,
                  float* C, int in_act, float slope, int act, const float* R, const float* R2, float div) {
  GemmArgs a;
  a.A = A; a.lda = Cin; a.W = cw.w; a.bias = cw.b; a.C = C; a.ldc = Cout; a.R = R; a.ldr = Cout; a.R2 = R2; a.ldr2 = Cout;
  a.M = T; a.N = Cout; a.Cin = Cin; a.taps = k; a.dil = dil; a.stride = 1; a.pad = dil * (k - 1) / 2; a.in_len = T;
  a.in_act = in_act; a.in_slope = slope; a.act = act; a.div = div; a.same_rows = 1;
  return launch_conv_gemm(a, s);
}

extern "C" int ss_vocoder_forward(ss_vocoder* v, void* stream, const int32_t* d_codes, int K, int dur_prediction,
                                  const int32_t* d_forced_dur, float* d_wav, int64_t wav_capacity,
                                  int32_t* d_dur, int64_t* h_n_samples) {
  if (!v || K <= 0 || !d_codes || !d_wav || !d_dur) return SS_ERR_ARG;
  SkScope sk_scope(v->skws);
  hipStream_t s = (hipStream_t)stream;
  const ss_vocoder_config& c = v->cfg;
  const int E = c.embedding_dim, Hd = c.dur_hidden;
  // --- embedding + duration predictor (codehifigan.py:56-66, fastspeech2.py:117-151) ---
  RET(v->small.ensure(((size_t)K * (E + 2 * Hd + 1) + 2 * (K + 2)) * sizeof(float)));
  float* emb = v->small.f();
  float* t1 = emb + (size_t)K * E;
  float* t2 = t1 + (size_t)K * Hd;
  float* logdur = t2 + (size_t)K * Hd;
  int* cum = reinterpret_cast<int*>(logdur + K);
  int* ones = cum + K + 1;
  RET(launch_gather_rows(d_codes, v->dict, E, emb, K, s, v->cfg.num_embeddings));
  const int* forced = d_forced_dur;
  if (!forced && dur_prediction) {
    RET(conv1d(s, emb, K, E, v->dur_c1, Hd, c.dur_kernel, 1, t1, ACT_NONE, 0.f, ACT_RELU, nullptr, nullptr, 0.f));
    RET(launch_layernorm(t1, Hd, t1, Hd, v->dur_ln1.g, v->dur_ln1.b, K, Hd, 1e-5f, s));
    RET(conv1d(s, t1, K, Hd, v->dur_c2, Hd, c.dur_kernel, 1, t2, ACT_NONE, 0.f, ACT_RELU, nullptr, nullptr, 0.f));
    RET(launch_layernorm(t2, Hd, t2, Hd, v->dur_ln2.g, v->dur_ln2.b, K, Hd, 1e-5f, s));
    RET(conv1d(s, t2, K, Hd, v->dur_proj, 1, 1, 1, logdur, ACT_NONE, 0.f, ACT_NONE, nullptr, nullptr, 0.f));
  } else if (!forced) {
    SS_HIP_CHECK(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(ones), 1, K, s));   // every unit lasts one frame
    forced = ones;
  }
  RET(launch_dur_predict(logdur, forced, K, d_dur, cum, s));
  int total = 0;
  SS_HIP_CHECK(hipMemcpyAsync(&total, cum + K, sizeof(int), hipMemcpyDeviceToHost, s));
  SS_HIP_CHECK(hipStreamSynchronize(s));
  const int Fr = total;
  int hop = 1;
  for (int i = 0; i < c.n_up; ++i) hop *= c.upsample_rates[i];
  const int64_t S = (int64_t)Fr * hop;
  if (h_n_samples) *h_n_samples = S;
  if (S > wav_capacity) return SS_ERR_CAPACITY;
  if (Fr <= 0) return SS_OK;

  // --- generator (hifigan.py:154-170).  Every stage holds T_i * C_i = Fr * hop_i * C0 / 2^(i+1) floats.
  size_t stage_max = (size_t)Fr * c.upsample_initial_channel;  // conv_pre output
  {
    int T = Fr, C = c.upsample_initial_channel;
    for (int i = 0; i < c.n_up; ++i) { T *= c.upsample_rates[i]; C /= 2; stage_max = std::max(stage_max, (size_t)T * C); }
  }
  RET(v->ws.ensure((8 * stage_max + (size_t)Fr * E) * sizeof(float)));
  float* frames = v->ws.f();
  GenBufs gb;
  gb.bx = frames + (size_t)Fr * E;       // stage input / MRF accumulator
  gb.bt = gb.bx + stage_max;             // conv1 output
  gb.br = gb.bt + stage_max;             // running resblock state
  gb.bs = gb.br + stage_max;             // x after the transposed conv
  gb.bxa = gb.bs + stage_max;            // leaky_relu twins of bx / br / bs (MFMA-bound stages only)
  gb.bra = gb.bxa + stage_max;
  gb.bsa = gb.bra + stage_max;
  gb.br2 = gb.bsa + stage_max;           // second resblock state (fused pairs ping-pong br / br2)
  RET(launch_repeat_rows(emb, cum, K, E, frames, Fr, s));
  int T = 1, C = 0;
  RET(hifigan_stack(v, s, [&](GemmArgs& a, int scale) { a.M = Fr * scale; a.in_len = Fr * scale; return launch_conv_gemm(a, s); },
                    [](int) { return SS_OK; },
                    [&](int scale, int& M, const int*& segs, int& nseg) { M = Fr * scale; segs = nullptr; nseg = 0; },
                    frames, Fr, gb, &T, &C));
  T *= Fr;
  float* bx = gb.bx;
  // leaky_relu (default slope 0.01, hifigan.py:166) -> conv_post -> tanh
  return launch_conv_post_tanh(bx, T, C, v->post.w, v->post.b, 0.01f, d_wav, s);
}


// =================================================================================================
// Ragged-batch stage twins: B independent utterances packed along the row axis.  No padding exists
// anywhere -- every utterance keeps the B = 1 arithmetic of the single-utterance entry points
// (SURVEY.md H2b); only launches, weight streaming and tile occupancy are shared.
// =================================================================================================
namespace {

// small int tables for the kernels: pageable -> device copies are staged by the runtime before
// hipMemcpyAsync returns, so the std::vector may die right after the call
int upload(hipStream_t s, int* dst, const std::vector<int>& v) {
  SS_HIP_CHECK(hipMemcpyAsync(dst, v.data(), v.size() * sizeof(int), hipMemcpyHostToDevice, s));
  return SS_OK;
}

struct Offsets { std::vector<int> off; int total = 0, mx = 0; };
Offsets prefix(const int* len, int B) {
  Offsets o; o.off.resize(B + 1); o.off[0] = 0;
  for (int b = 0; b < B; ++b) { o.off[b + 1] = o.off[b] + len[b]; o.mx = std::max(o.mx, len[b]); }
  o.total = o.off[B];
  return o;
}

}  // namespace

extern "C" int ss_batch_fbank_cmvn(ss_model* m, void* stream, int B, const float* d_pcm, const int64_t* h_pcm_start,
                                   const int32_t* h_n_samples, float pcm_scale, float* d_feat, int32_t* h_T) {
  if (!m || B <= 0) return SS_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  std::vector<int> segs(3 * B);
  int row = 0, mx = 0;
  for (int b = 0; b < B; ++b) {
    const int T = ss_fbank_num_frames(h_n_samples[b]);
    h_T[b] = T;
    segs[3 * b] = (int)h_pcm_start[b]; segs[3 * b + 1] = T; segs[3 * b + 2] = row;
    row += T; mx = std::max(mx, T);
  }
  RET(m->seg_buf.ensure(segs.size() * sizeof(int)));
  RET(upload(s, (int*)m->seg_buf.p, segs));
  return launch_fbank_cmvn_batch(d_pcm, pcm_scale, m->fe_window, m->fe_melw, m->fe_mean, m->fe_std, d_feat,
                                 (const int*)m->seg_buf.p, B, mx, s);
}

extern "C" int ss_batch_encoder_forward(ss_model* m, void* stream, int B, const float* d_fbank, const int32_t* h_T,
                                        int attn_chunk, int conv_chunk, float* d_enc_out, int32_t* h_Tp) {
  if (!m || B <= 0) return SS_ERR_ARG;
  SkScope sk_scope(m->skws);
  CanonScope canon_scope(m->pack_invariant ? CANON_SEQ : CANON_NONE);
  const bool canon = m->pack_invariant && !debug_tile_forced();
  hipStream_t s = (hipStream_t)stream;
  const ss_config& c = m->cfg;
  const int d = c.enc_dim, f = c.enc_ffn, k = c.conv_kernel, Ld = c.enc_layers * d;
  std::vector<int> T1(B), T2(B);
  for (int b = 0; b < B; ++b) {
    if (h_T[b] <= 0) return SS_ERR_ARG;
    T1[b] = conv_out_len(h_T[b], k, 2); T2[b] = conv_out_len(T1[b], k, 2);
    if (T2[b] <= 0 || T2[b] > c.max_rel_pos) return SS_ERR_CAPACITY;
    h_Tp[b] = T2[b];
  }
  const Offsets o0 = prefix(h_T, B), o1 = prefix(T1.data(), B), o2 = prefix(T2.data(), B);
  const int M1 = o1.total, M2 = o2.total;
  const int cchunk = (conv_chunk > 0 && conv_chunk < 999) ? conv_chunk : 0;
  const int achunk = attn_chunk > 0 && attn_chunk < 999999 ? attn_chunk : 0;

  // segment tables: conv0 {out,in}, conv1 {out,in}, attention {q,k}, rows {start,len}
  std::vector<int> tab(14 * B);
  int* t0 = tab.data(); int* t1 = t0 + 4 * B; int* ta = t1 + 4 * B; int* tr = ta + 4 * B;
  for (int b = 0; b < B; ++b) {
    t0[4 * b] = o1.off[b]; t0[4 * b + 1] = T1[b]; t0[4 * b + 2] = o0.off[b]; t0[4 * b + 3] = h_T[b];
    t1[4 * b] = o2.off[b]; t1[4 * b + 1] = T2[b]; t1[4 * b + 2] = o1.off[b]; t1[4 * b + 3] = T1[b];
    ta[4 * b] = o2.off[b]; ta[4 * b + 1] = T2[b]; ta[4 * b + 2] = o2.off[b]; ta[4 * b + 3] = T2[b];
    tr[2 * b] = o2.off[b]; tr[2 * b + 1] = T2[b];
  }
  RET(m->seg_buf.ensure(tab.size() * sizeof(int)));
  int* dt = (int*)m->seg_buf.p;
  RET(upload(s, dt, tab));
  const int *d0 = dt, *d1 = dt + 4 * B, *da = dt + 8 * B, *dr = dt + 12 * B;

  const size_t n_h1 = (size_t)M1 * (c.conv_channels / 2), n_x = (size_t)M2 * d, n_f = (size_t)M2 * f,
               n_qkv = (size_t)M2 * 3 * d;
  RET(m->ws.ensure((n_h1 + 3 * n_x + n_f + n_qkv) * sizeof(float)));
  float* h1 = m->ws.f();
  float* x = d_enc_out;
  float* h = h1 + n_h1;
  float* g = h + n_x;
  float* g2 = g + n_x;
  float* ff = g2 + n_x;
  float* qkv = ff + n_f;
  {
    GemmArgs a;
    a.A = d_fbank; a.lda = c.input_feat; a.W = m->sub0.w; a.bias = m->sub0.b; a.C = h1; a.ldc = c.conv_channels / 2;
    a.N = c.conv_channels; a.Cin = c.input_feat; a.taps = k; a.stride = 2; a.pad = k / 2; a.chunk = cchunk; a.glu = 1;
    a.segs = d0; a.nseg = B; a.max_seg_out = o1.mx; a.M = M1; a.in_len = o0.total;
    RET(launch_conv_gemm(a, s));
    GemmArgs b2;
    b2.A = h1; b2.lda = c.conv_channels / 2; b2.W = m->sub1.w; b2.bias = m->sub1.b; b2.C = g; b2.ldc = d;
    b2.N = 2 * d; b2.Cin = c.conv_channels / 2; b2.taps = k; b2.stride = 2; b2.pad = k / 2; b2.chunk = cchunk; b2.glu = 1;
    b2.segs = d1; b2.nseg = B; b2.max_seg_out = o2.mx; b2.M = M2; b2.in_len = M1;
    RET(launch_conv_gemm(b2, s));
  }
  RET(linear(s, g, d, M2, m->enc_linear, d, d, x, d));
  for (int l = 0; l < c.enc_layers; ++l) {
    const EncLayer& e = m->enc[l];
    // macaron FFN: x += 0.5 * W2 SiLU(W1 LN(x)); packed batches: ONE launch (ffn.hip), the [rows, 2048] hidden tile stays on chip
    // (pack-invariant contexts: ALWAYS the fused launch in its whole-tile form -- the two-launch form sums the 2048 hidden terms in
    //  another order, and which of the two runs must not depend on the row count)
    const bool fuse_ffn = (canon || (disp().ffn_fusion && M2 >= disp().ffn_min_rows)) && ffn_fused_eligible(d, f, ACT_SILU, M2, d, d) &&
                          e.ffn1_w1.b && e.ffn1_w2.b && e.ffn2_w1.b && e.ffn2_w2.b;
    if (fuse_ffn) {
      RET(launch_ffn_fused(x, d, x, d, e.ffn1_ln.g, e.ffn1_ln.b, e.ffn1_w1.w, e.ffn1_w1.b, e.ffn1_w2.w, e.ffn1_w2.b, 0.5f, nullptr,
                           nullptr, M2, d, f, s, canon));
    } else {
      RET(ln_linear(s, x, M2, e.ffn1_ln, e.ffn1_w1, f, d, ff, f, h, ACT_SILU));
      RET(linear(s, ff, f, M2, e.ffn1_w2, d, f, x, d, ACT_NONE, 0.5f, x, d));
    }
    RET(ln_linear(s, x, M2, e.attn_ln, e.qkv, 3 * d, d, qkv, 3 * d, h));
    AttnArgs at;
    at.Q = qkv; at.K = qkv + d; at.V = qkv + 2 * d; at.ldq = at.ldk = at.ldv = 3 * d;
    at.O = h; at.ldo = d; at.H = c.enc_heads; at.scale = 0.125f; at.chunk = achunk;
    at.P = m->pos_proj.f() + (size_t)l * d; at.ldp = Ld; at.p_tmax = c.max_rel_pos; at.bias_u = e.u; at.bias_v = e.v;
    at.segs = da; at.nseg = B; at.max_q = o2.mx;
    RET(launch_attention(at, s));
    RET(linear(s, h, d, M2, e.out, d, d, x, d, ACT_NONE, 1.f, x, d));
    RET(ln_linear(s, x, M2, e.conv_ln, e.pw1, 2 * d, d, g, d, h, ACT_NONE, 1.f, 1));
    RET(launch_dwconv_bn_silu(g, d, g2, d, e.dw_wt, c.dw_kernel, e.bn_mean, e.bn_var, e.bn_g, e.bn_b, 1e-5f,
                              o2.mx, d, cchunk, s, dr, B));
    RET(linear(s, g2, d, M2, e.pw2, d, d, x, d, ACT_NONE, 1.f, x, d));
    if (fuse_ffn) {                          // second FFN + the layer's final LayerNorm in the same launch
      RET(launch_ffn_fused(x, d, x, d, e.ffn2_ln.g, e.ffn2_ln.b, e.ffn2_w1.w, e.ffn2_w1.b, e.ffn2_w2.w, e.ffn2_w2.b, 0.5f,
                           e.final_ln.g, e.final_ln.b, M2, d, f, s, canon));
    } else {
      RET(ln_linear(s, x, M2, e.ffn2_ln, e.ffn2_w1, f, d, ff, f, h, ACT_SILU));
      RET(linear(s, ff, f, M2, e.ffn2_w2, d, f, x, d, ACT_NONE, 0.5f, x, d));
      RET(layernorm(s, x, x, e.final_ln, M2, d));
    }
  }
  return SS_OK;
}

extern "C" int ss_batch_ctc_greedy(ss_model* m, void* stream, int head, int B, const float* d_enc_out,
                                   const int32_t* h_Tp, int32_t* d_raw, int32_t* d_tokens, int32_t* d_index,
                                   int32_t* d_counts) {
  if (!m || B <= 0 || head < 0 || head > 1) return SS_ERR_ARG;
  SkScope sk_scope(m->skws);
  CanonScope canon_scope(m->pack_invariant ? CANON_SEQ : CANON_NONE);
  hipStream_t s = (hipStream_t)stream;
  const ss_config& c = m->cfg;
  const Offsets o = prefix(h_Tp, B);
  const int V = head == 0 ? c.src_vocab : c.tgt_vocab;
  RET(m->mt_ws.ensure((size_t)o.total * V * sizeof(float)));
  float* logits = m->mt_ws.f();
  std::vector<int> tr(2 * B);
  for (int b = 0; b < B; ++b) { tr[2 * b] = o.off[b]; tr[2 * b + 1] = h_Tp[b]; }
  RET(m->seg_buf.ensure(tr.size() * sizeof(int)));
  RET(upload(s, (int*)m->seg_buf.p, tr));
  RET(linear(s, d_enc_out, c.enc_dim, o.total, head == 0 ? m->ctc_asr : m->ctc_st, V, c.enc_dim, logits, V));
  m->dbg_logits = logits; m->dbg_rows = o.total; m->dbg_cols = V;
  RET(launch_masked_argmax(logits, V, o.total, V, c.pad, c.unk, -1, -1, d_raw, s));
  return launch_ctc_collapse(d_raw, 0, 0, c.pad, d_tokens, d_index, d_counts, s, (const int*)m->seg_buf.p, B);
}

// Batched beam-1 search: all utterances start from [</s>] and advance in lockstep, one row per
// utterance (M = B GEMMs stream every decoder weight once per step for the whole batch).
extern "C" int ss_batch_mt_greedy(ss_model* m, void* stream, int B, const float* d_enc_out, const int32_t* h_Tp,
                                  const int32_t* h_max_len, int min_len, int32_t* h_out_tokens, int out_stride,
                                  int32_t* h_n_out, float* d_feats, int feat_rows) {
  if (!m || B <= 0 || B > 128 || !d_feats) return SS_ERR_ARG;
  SkScope sk_scope(m->skws);
  CanonScope canon_scope(m->pack_invariant ? CANON_SEQ : CANON_NONE);     // cross K|V over the packed encoder rows
  hipStream_t s = (hipStream_t)stream;
  const ss_config& c = m->cfg;
  const int D = c.dec_dim, F = c.dec_ffn, V = c.tgt_vocab, H = c.dec_heads;
  int Lmax = 0;
  for (int b = 0; b < B; ++b) {
    if (h_Tp[b] <= 0 || h_max_len[b] < 0) return SS_ERR_ARG;   // an utterance without encoder rows has nothing to attend to
    Lmax = std::max(Lmax, h_max_len[b]);
  }
  const int Lcap = Lmax + 2;
  if (Lcap > feat_rows || Lcap + 2 > c.max_tgt_pos || out_stride < Lmax + 1) return SS_ERR_CAPACITY;
  const Offsets oe = prefix(h_Tp, B);
  // cross-attention K/V for every layer over the packed encoder rows
  RET(m->mt_cross.ensure((size_t)c.mt_layers * oe.total * 2 * D * sizeof(float)));
  for (int l = 0; l < c.mt_layers; ++l)
    RET(linear(s, d_enc_out, c.enc_dim, oe.total, m->mt[l].cross_kv, 2 * D, c.enc_dim,
               m->mt_cross.f() + (size_t)l * oe.total * 2 * D, 2 * D));
  // caches / scratch
  RET(m->bmt_self.ensure((size_t)c.mt_layers * B * Lcap * 3 * D * sizeof(float)));
  RET(m->mt_ws.ensure(((size_t)B * (3 * D + F + V)) * sizeof(float)));
  float* x = m->mt_ws.f();
  float* h = x + (size_t)B * D;
  float* q2 = h + (size_t)B * D;
  float* ff = q2 + (size_t)B * D;
  float* logits = ff + (size_t)B * F;
  // int tables: tokens [Lcap+1][B], max_len [B], cross segs [B][4], self segs per step [Lcap][B][4]
  const size_t n_tok = (size_t)(Lcap + 1) * B;
  RET(m->seg_buf.ensure((n_tok + B + 4 * B + (size_t)Lcap * 4 * B) * sizeof(int)));
  int* tok = (int*)m->seg_buf.p;
  int* d_maxlen = tok + n_tok;
  int* d_cross = d_maxlen + B;
  int* d_self = d_cross + 4 * B;
  {
    std::vector<int> t0(B, c.eos), ml(h_max_len, h_max_len + B), cs(4 * B), ss((size_t)Lcap * 4 * B);
    for (int b = 0; b < B; ++b) { cs[4 * b] = b; cs[4 * b + 1] = 1; cs[4 * b + 2] = oe.off[b]; cs[4 * b + 3] = h_Tp[b]; }
    for (int st = 0; st < Lcap; ++st)
      for (int b = 0; b < B; ++b) {
        int* e = &ss[((size_t)st * B + b) * 4];
        e[0] = b; e[1] = 1; e[2] = b * Lcap; e[3] = st + 1;
      }
    RET(upload(s, tok, t0)); RET(upload(s, d_maxlen, ml)); RET(upload(s, d_cross, cs)); RET(upload(s, d_self, ss));
  }
  std::vector<int> host_tok(n_tok, c.pad);
  std::vector<int> eos_at(B, -1);
  int checked = 1;     // token rows [1, checked) already copied to the host
  int step = 0;        // position being fed
  constexpr int kCheck = 4;
  // the decode rows (one per utterance): the small-M kernel in a split-K form fixed by the layer shape -- not by B (with B <= 4 the
  // heuristic would take the GEMV, with B = 64 another wave arrangement for the vocabulary projection)
  CanonScope decode_scope(m->pack_invariant ? CANON_SMALLM : CANON_NONE);
  while (true) {
    // feed position `step` of every utterance
    RET(launch_embed_tokens(tok + (size_t)step * B, m->mt_emb, m->mt_pos, sqrtf((float)D), step + c.pad + 1, x, B, D, s, 0, -1, c.tgt_vocab));
    for (int l = 0; l < c.mt_layers; ++l) {
      float* cache = m->bmt_self.f() + (size_t)l * B * Lcap * 3 * D;
      float* rows = cache + (size_t)step * 3 * D;                    // row b at + b*Lcap*3D
      AttnArgs at;
      at.Q = rows; at.ldq = Lcap * 3 * D; at.K = cache + D; at.V = cache + 2 * D; at.ldk = at.ldv = 3 * D;
      at.O = h; at.ldo = D; at.H = H; at.scale = 1.f; at.causal = 0;   // cache holds exactly the visible keys
      at.segs = d_self + (size_t)step * 4 * B; at.nseg = B; at.max_q = 1;
      AttnArgs ac;
      ac.Q = q2; ac.ldq = D; ac.K = m->mt_cross.f() + (size_t)l * oe.total * 2 * D; ac.V = ac.K + D; ac.ldk = ac.ldv = 2 * D;
      ac.O = h; ac.ldo = D; ac.H = H; ac.scale = 1.f; ac.segs = d_cross; ac.nseg = B; ac.max_q = 1;
      RET(dec_layer_ex(s, c, m->mt[l], x, B, rows, Lcap * 3 * D, at, &ac, h, q2, ff));
    }
    float* frow = d_feats + (size_t)step * D;                         // utterance b at + b*feat_rows*D
    RET(launch_layernorm(x, D, frow, feat_rows * D, m->mt_ln.g, m->mt_ln.b, B, D, 1e-5f, s));
    Lin proj{m->mt_emb, nullptr};
    RET(linear(s, frow, feat_rows * D, B, proj, V, D, logits, V));
    RET(launch_masked_argmax(logits, V, B, V, c.pad, step < min_len ? c.eos : -1, -1, -1, tok + (size_t)(step + 1) * B, s,
                             d_maxlen, step, c.eos));
    ++step;                                                           // tokens of row `step` now exist
    const bool last = step > Lmax;
    if (last || step % kCheck == 0) {
      SS_HIP_CHECK(hipMemcpyAsync(host_tok.data() + (size_t)checked * B, tok + (size_t)checked * B,
                                  (size_t)(step + 1 - checked) * B * sizeof(int), hipMemcpyDeviceToHost, s));
      SS_HIP_CHECK(hipStreamSynchronize(s));
      bool all_done = true;
      for (int b = 0; b < B; ++b) {
        for (int r = checked; r <= step && eos_at[b] < 0; ++r)
          if (host_tok[(size_t)r * B + b] == c.eos) eos_at[b] = r;
        if (eos_at[b] < 0) all_done = false;
      }
      checked = step + 1;
      if (all_done || last) break;
    }
  }
  for (int b = 0; b < B; ++b) {
    const int end = eos_at[b] >= 0 ? eos_at[b] : step;      // row of the last generated token
    h_n_out[b] = end;                                        // tokens generated = rows 1..end
    for (int r = 1; r <= end; ++r) h_out_tokens[(size_t)b * out_stride + (r - 1)] = host_tok[(size_t)r * B + b];
  }
  return SS_OK;
}

extern "C" int ss_batch_t2u_units(ss_model* m, void* stream, int B, const float* d_feats, int feat_rows,
                                  const int32_t* h_n, int t2u_causal, int mask_eos, int32_t* d_raw, int32_t* d_tokens,
                                  int32_t* d_counts) {
  if (!m || B <= 0) return SS_ERR_ARG;
  SkScope sk_scope(m->skws);
  CanonScope canon_scope(m->pack_invariant ? CANON_SEQ : CANON_NONE);
  hipStream_t s = (hipStream_t)stream;
  const ss_config& c = m->cfg;
  const int D = c.dec_dim, F = c.dec_ffn, V = c.unit_vocab, H = c.dec_heads, up = c.ctc_upsample;
  for (int b = 0; b < B; ++b)
    if (h_n[b] <= 0) return SS_ERR_ARG;                    // every utterance feeds at least the leading </s> state
  const Offsets on = prefix(h_n, B);
  const int Nn = on.total, U = Nn * up;
  const size_t nx = (size_t)U * D;
  RET(m->ws.ensure((3 * nx + (size_t)U * 3 * D + (size_t)U * F + (size_t)Nn * 2 * D + (size_t)Nn * D +
                    (size_t)U * V + (size_t)U) * sizeof(float)));
  float* x = m->ws.f();
  float* h = x + nx;
  float* q2 = h + nx;
  float* selfbuf = q2 + nx;
  float* ff = selfbuf + (size_t)U * 3 * D;
  float* crosskv = ff + (size_t)U * F;
  float* t2u_out = crosskv + (size_t)Nn * 2 * D;
  float* logits = t2u_out + (size_t)Nn * D;
  int32_t* idx_scratch = reinterpret_cast<int32_t*>(logits + (size_t)U * V);
  // tables: t2u self {off,n,off,n}; unit self {25off,25n,25off,25n}; unit cross {25off,25n,off,n}; rows {25off,25n}
  std::vector<int> tab(14 * B);
  for (int b = 0; b < B; ++b) {
    const int o = on.off[b], n = h_n[b];
    int* a = &tab[4 * b]; a[0] = o; a[1] = n; a[2] = o; a[3] = n;
    int* u = &tab[4 * B + 4 * b]; u[0] = o * up; u[1] = n * up; u[2] = o * up; u[3] = n * up;
    int* x2 = &tab[8 * B + 4 * b]; x2[0] = o * up; x2[1] = n * up; x2[2] = o; x2[3] = n;
    tab[12 * B + 2 * b] = o * up; tab[12 * B + 2 * b + 1] = n * up;
  }
  RET(m->seg_buf.ensure(tab.size() * sizeof(int)));
  int* dt = (int*)m->seg_buf.p;
  RET(upload(s, dt, tab));
  // gather the decoder states of each utterance into packed rows
  for (int b = 0; b < B; ++b)
    SS_HIP_CHECK(hipMemcpyAsync(x + (size_t)on.off[b] * D, d_feats + (size_t)b * feat_rows * D,
                                (size_t)h_n[b] * D * sizeof(float), hipMemcpyDeviceToDevice, s));
  for (int l = 0; l < c.t2u_layers; ++l) {
    AttnArgs at;
    at.Q = selfbuf; at.ldq = 3 * D; at.K = selfbuf + D; at.V = selfbuf + 2 * D; at.ldk = at.ldv = 3 * D;
    at.O = h; at.ldo = D; at.H = H; at.scale = 1.f; at.causal = t2u_causal ? 1 : 0;
    at.segs = dt; at.nseg = B; at.max_q = on.mx;
    at.no_decode_kernel = m->pack_invariant;       // max_q is the pack's longest utterance: it must not pick the kernel
    RET(dec_layer_ex(s, c, m->t2u[l], x, Nn, selfbuf, 3 * D, at, nullptr, h, q2, ff));
  }
  RET(launch_layernorm(x, D, t2u_out, D, m->t2u_ln.g, m->t2u_ln.b, Nn, D, 1e-5f, s));
  RET(launch_upsample_add_pos(t2u_out, Nn, up, m->unit_pos_row, (float)c.pad, x, D, s));
  for (int l = 0; l < c.unit_layers; ++l) {
    RET(linear(s, t2u_out, D, Nn, m->unit[l].cross_kv, 2 * D, D, crosskv, 2 * D));
    AttnArgs at;
    at.Q = selfbuf; at.ldq = 3 * D; at.K = selfbuf + D; at.V = selfbuf + 2 * D; at.ldk = at.ldv = 3 * D;
    at.O = h; at.ldo = D; at.H = H; at.scale = 1.f; at.causal = 1;
    at.segs = dt + 4 * B; at.nseg = B; at.max_q = on.mx * up; at.no_decode_kernel = m->pack_invariant;
    AttnArgs ac;
    ac.Q = q2; ac.ldq = D; ac.K = crosskv; ac.V = crosskv + D; ac.ldk = ac.ldv = 2 * D;
    ac.O = h; ac.ldo = D; ac.H = H; ac.scale = 1.f; ac.segs = dt + 8 * B; ac.nseg = B; ac.max_q = on.mx * up;
    ac.no_decode_kernel = m->pack_invariant;
    RET(dec_layer_ex(s, c, m->unit[l], x, U, selfbuf, 3 * D, at, &ac, h, q2, ff));
  }
  RET(launch_layernorm(x, D, h, D, m->unit_ln.g, m->unit_ln.b, U, D, 1e-5f, s));
  RET(linear(s, h, D, U, m->unit_out, V, D, logits, V));
  m->dbg_logits = logits; m->dbg_rows = U; m->dbg_cols = V;
  RET(launch_masked_argmax(logits, V, U, V, c.pad, c.unk, mask_eos ? c.eos : -1, -1, d_raw, s));
  return launch_ctc_collapse(d_raw, 0, V - 1, c.pad, d_tokens, idx_scratch, d_counts, s, dt + 12 * B, B);
}

extern "C" int ss_batch_vocoder_forward(ss_vocoder* v, void* stream, int B, const int32_t* d_codes, const int32_t* h_K,
                                        int dur_prediction, const int32_t* d_forced_dur, float* d_wav,
                                        int64_t wav_capacity, int32_t* d_dur, int64_t* h_wav_start,
                                        int64_t* h_n_samples) {
  if (!v || B <= 0 || !d_codes || !d_wav || !d_dur) return SS_ERR_ARG;
  SkScope sk_scope(v->skws);
  hipStream_t s = (hipStream_t)stream;
  const ss_vocoder_config& c = v->cfg;
  const int E = c.embedding_dim, Hd = c.dur_hidden;
  for (int b = 0; b < B; ++b)
    if (h_K[b] <= 0) return SS_ERR_ARG;                    // the single-utterance form refuses K = 0 too; callers drop unit-less utterances
  const Offsets ok = prefix(h_K, B);
  const int Kt = ok.total;
  RET(v->small.ensure(((size_t)Kt * (E + 2 * Hd + 1) + 2 * (Kt + B + 2)) * sizeof(float)));
  float* emb = v->small.f();
  float* t1 = emb + (size_t)Kt * E;
  float* t2 = t1 + (size_t)Kt * Hd;
  float* logdur = t2 + (size_t)Kt * Hd;
  int* cum = reinterpret_cast<int*>(logdur + Kt);           // Kt + B entries (one extra per utterance)
  int* ones = cum + Kt + B + 1;
  // unit-axis tables: conv segs {out,len,in,len} and {start,len}
  std::vector<int> tk(6 * B);
  for (int b = 0; b < B; ++b) {
    int* a = &tk[4 * b]; a[0] = ok.off[b]; a[1] = h_K[b]; a[2] = ok.off[b]; a[3] = h_K[b];
    tk[4 * B + 2 * b] = ok.off[b]; tk[4 * B + 2 * b + 1] = h_K[b];
  }
  RET(v->segs.ensure((6 * B + 16 * B) * sizeof(int)));
  int* dk = (int*)v->segs.p;
  RET(upload(s, dk, tk));
  RET(launch_gather_rows(d_codes, v->dict, E, emb, Kt, s, v->cfg.num_embeddings));
  const int* forced = d_forced_dur;
  auto sconv = [&](const float* A, int Cin, const ConvW& cw, int Cout, int k, float* Cc, int act) {
    GemmArgs a;
    a.A = A; a.lda = Cin; a.W = cw.w; a.bias = cw.b; a.C = Cc; a.ldc = Cout; a.N = Cout; a.Cin = Cin; a.taps = k;
    a.pad = (k - 1) / 2; a.act = act; a.segs = dk; a.nseg = B; a.max_seg_out = ok.mx; a.M = Kt; a.in_len = Kt;
    return launch_conv_gemm(a, s);
  };
  if (!forced && dur_prediction) {
    RET(sconv(emb, E, v->dur_c1, Hd, c.dur_kernel, t1, ACT_RELU));
    RET(launch_layernorm(t1, Hd, t1, Hd, v->dur_ln1.g, v->dur_ln1.b, Kt, Hd, 1e-5f, s));
    RET(sconv(t1, Hd, v->dur_c2, Hd, c.dur_kernel, t2, ACT_RELU));
    RET(launch_layernorm(t2, Hd, t2, Hd, v->dur_ln2.g, v->dur_ln2.b, Kt, Hd, 1e-5f, s));
    RET(sconv(t2, Hd, v->dur_proj, 1, 1, logdur, ACT_NONE));
  } else if (!forced) {
    std::vector<int> one(Kt, 1);
    RET(upload(s, ones, one));
    forced = ones;
  }
  RET(launch_dur_predict(logdur, forced, 0, d_dur, cum, s, dk + 4 * B, B));
  std::vector<int> hcum(Kt + B);
  SS_HIP_CHECK(hipMemcpyAsync(hcum.data(), cum, (size_t)(Kt + B) * sizeof(int), hipMemcpyDeviceToHost, s));
  SS_HIP_CHECK(hipStreamSynchronize(s));
  int hop = 1;
  for (int i = 0; i < c.n_up; ++i) hop *= c.upsample_rates[i];
  std::vector<int> Fr(B);
  for (int b = 0; b < B; ++b) Fr[b] = hcum[ok.off[b] + b + h_K[b]];
  const Offsets of = prefix(Fr.data(), B);
  for (int b = 0; b < B; ++b) { h_wav_start[b] = (int64_t)of.off[b] * hop; h_n_samples[b] = (int64_t)Fr[b] * hop; }
  if ((int64_t)of.total * hop > wav_capacity) return SS_ERR_CAPACITY;
  if (of.total <= 0) return SS_OK;
  const int Ft = of.total;

  size_t stage_max = (size_t)Ft * c.upsample_initial_channel;
  {
    int T = Ft, C = c.upsample_initial_channel;
    for (int i = 0; i < c.n_up; ++i) { T *= c.upsample_rates[i]; C /= 2; stage_max = std::max(stage_max, (size_t)T * C); }
  }
  RET(v->ws.ensure((8 * stage_max + (size_t)Ft * E) * sizeof(float)));
  float* frames = v->ws.f();
  GenBufs gb;
  gb.bx = frames + (size_t)Ft * E;
  gb.bt = gb.bx + stage_max;
  gb.br = gb.bt + stage_max;
  gb.bs = gb.br + stage_max;
  gb.bxa = gb.bs + stage_max;
  gb.bra = gb.bxa + stage_max;
  gb.bsa = gb.bra + stage_max;
  gb.br2 = gb.bsa + stage_max;
  // frame-axis tables, rebuilt per stage (rows scale by the running hop)
  int* dseg = dk + 6 * B;            // conv segs [B][4]
  int* drep = dseg + 4 * B;          // repeat_rows segs [B][4]
  int* dwav = drep + 4 * B;          // conv_post segs [B][2]
  auto stage_segs = [&](int scale) {
    std::vector<int> t(4 * B);
    for (int b = 0; b < B; ++b) { t[4 * b] = of.off[b] * scale; t[4 * b + 1] = Fr[b] * scale; t[4 * b + 2] = t[4 * b]; t[4 * b + 3] = t[4 * b + 1]; }
    return upload(s, dseg, t);
  };
  {
    std::vector<int> t(4 * B);
    for (int b = 0; b < B; ++b) { t[4 * b] = ok.off[b]; t[4 * b + 1] = h_K[b]; t[4 * b + 2] = of.off[b]; t[4 * b + 3] = Fr[b]; }
    RET(upload(s, drep, t));
  }
  RET(launch_repeat_rows(emb, cum, 0, E, frames, of.mx, s, drep, B));
  int scale = 1, C = 0;
  RET(hifigan_stack(v, s,
                    [&](GemmArgs& a, int sc) {
                      a.segs = dseg; a.nseg = B; a.max_seg_out = of.mx * sc; a.M = Ft * sc; a.in_len = Ft * sc;
                      return launch_conv_gemm(a, s);
                    },
                    stage_segs,
                    [&](int sc, int& M, const int*& segs, int& nseg) { M = Ft * sc; segs = dseg; nseg = B; },
                    frames, Ft, gb, &scale, &C));
  float* bx = gb.bx;
  {
    std::vector<int> t(2 * B);
    for (int b = 0; b < B; ++b) { t[2 * b] = of.off[b] * scale; t[2 * b + 1] = Fr[b] * scale; }
    RET(upload(s, dwav, t));
  }
  return launch_conv_post_tanh(bx, of.mx * scale, C, v->post.w, v->post.b, 0.01f, d_wav, s, dwav, B);
}

// =================================================================================================
// op-level entry points
// =================================================================================================
extern "C" int ss_op_conv_gemm(void* stream, const float* dA, int lda, const float* dW, const float* dbias,
                               const float* dR, int ldr, const float* dR2, int ldr2, float* dC, int ldc, int M,
                               int N, int Cin, int taps, int dil, int stride, int pad, int in_len, int chunk,
                               int in_act, float in_slope, int act, float alpha, float div, int glu) {
  GemmArgs a;
  a.A = dA; a.lda = lda; a.W = dW; a.bias = dbias; a.R = dR; a.ldr = ldr; a.R2 = dR2; a.ldr2 = ldr2; a.C = dC; a.ldc = ldc;
  a.M = M; a.N = N; a.Cin = Cin; a.taps = taps; a.dil = dil; a.stride = stride; a.pad = pad; a.in_len = in_len;
  a.chunk = chunk; a.in_act = in_act; a.in_slope = in_slope; a.act = act; a.alpha = alpha; a.div = div; a.glu = glu;
  a.same_rows = (stride == 1 && M == in_len) ? 1 : 0;
  // unit-test path of the Winograd form (the model makes the transformed weights once per context): made here per call
  // (one buffer per stream of the calling thread: a re-pack for a launch on stream B must not overwrite the weights a kernel
  //  queued on stream A is still reading -- ADVICE r4)
  static thread_local std::map<hipStream_t, DevBuf> wino_tmps;
  DevBuf& wino_tmp = wino_tmps[(hipStream_t)stream];
  if ((conv_c64w_enabled() && N == 64 && Cin == 64 && taps >= 3 && conv_c64_eligible(a)) ||
      (conv_c128w_enabled() && N == 128 && Cin == 128 && taps >= 3 && a.same_rows && !glu) ||
      (conv_c256w_enabled() && N == 256 && Cin == 256 && taps >= 3 && a.same_rows && !glu) ||
      (conv_c32w_enabled() && N == 32 && Cin == 32 && taps >= 3 && conv_c32_eligible(a))) {
    RET(wino_tmp.ensure((size_t)N * ((taps + 2) / 3) * 4 * N * sizeof(float)));
    RET(launch_wino_pack(dW, wino_tmp.f(), N, taps, (hipStream_t)stream));
    a.Wwino = wino_tmp.f();
  }
  return launch_conv_gemm(a, (hipStream_t)stream);
}

extern "C" int ss_model_set_pack_invariant(ss_model* m, int on) {
  if (!m) return SS_ERR_ARG;
  m->pack_invariant = on ? 1 : 0;
  return SS_OK;
}
extern "C" int ss_model_get_pack_invariant(ss_model* m) { return m ? m->pack_invariant : SS_ERR_ARG; }

extern "C" int ss_debug_last_logits(ss_model* m, void* stream, float* d_out, int64_t cap_floats, int* h_rows, int* h_cols) {
  if (!m || !h_rows || !h_cols) return SS_ERR_ARG;
  *h_rows = m->dbg_rows; *h_cols = m->dbg_cols;
  if (!d_out) return SS_OK;                                   // size query
  if (!m->dbg_logits || cap_floats < (int64_t)m->dbg_rows * m->dbg_cols) return SS_ERR_CAPACITY;
  SS_HIP_CHECK(hipMemcpyAsync(d_out, m->dbg_logits, (size_t)m->dbg_rows * m->dbg_cols * sizeof(float), hipMemcpyDeviceToDevice,
                              (hipStream_t)stream));
  return SS_OK;
}

extern "C" int ss_op_ffn_fused(void* stream, const float* dX, int ldx, float* dY, int ldy, const float* ln_g, const float* ln_b,
                               const float* dW1, const float* db1, const float* dW2, const float* db2, float alpha,
                               const float* ln2_g, const float* ln2_b, int M, int D, int F) {
  return launch_ffn_fused(dX, ldx, dY, ldy, ln_g, ln_b, dW1, db1, dW2, db2, alpha, ln2_g, ln2_b, M, D, F, (hipStream_t)stream,
                          canon_mode() == CANON_SEQ);
}
// Test hook: the arithmetic mode of the ss_op_* entry points called from this thread (0 fastest kernel per shape, 1 the pack-invariant
// one-chain form, 2 the fixed small-M form of the lock-step decode rows); the model entry points set their own.
extern "C" int ss_debug_canon(int mode) {
  if (mode < 0 || mode > 2) return SS_ERR_ARG;
  canon_debug_set(mode);
  return SS_OK;
}
extern "C" int ss_op_ln_linear(void* stream, const float* dX, int ldx, const float* ln_g, const float* ln_b, const float* dW,
                               const float* dbias, const float* dR, int ldr, float* dC, int ldc, int M, int N, int K, int act,
                               float alpha, int glu) {
  GemmArgs a;
  a.A = dX; a.lda = ldx; a.W = dW; a.bias = dbias; a.R = dR; a.ldr = ldr; a.C = dC; a.ldc = ldc;
  a.M = M; a.N = N; a.Cin = K; a.in_len = M; a.act = act; a.alpha = alpha; a.glu = glu; a.same_rows = 1;
  a.ln_g = ln_g; a.ln_b = ln_b;
  return launch_conv_gemm(a, (hipStream_t)stream);       // SS_ERR_ARG when no kernel with a LayerNorm prologue takes the shape
}
// enable 0 / 1: the stage on conv_sk2<64> / on the slab kernels; 4 / 5: its Winograd form (conv_c64w.hip) off / on (the slab kernels stay on);
// 6 / 7: the 128-channel stage on conv_sk2<128> / on the Winograd slab kernel
extern "C" int ss_debug_conv_c64(int enable) {
  if (enable == 4 || enable == 5) { conv_c64w_debug(enable == 5); return SS_OK; }
  if (enable == 6 || enable == 7) { conv_c128w_debug(enable == 7); return SS_OK; }     // the 128-channel stage: conv_sk2<128> + twins / Winograd slab
  if (enable == 8 || enable == 9) { conv_c256w_debug(enable == 9); return SS_OK; }     // the 256-channel stage: conv_sk2<128> + twins / Winograd slab (two phases)
  conv_c64_debug(enable);
  return SS_OK;
}
extern "C" int ss_debug_conv_c32(int enable) {      // 0 / 1: the per-conv slab kernel off / on; 4 / 5: its Winograd form off / on
  if (enable == 4 || enable == 5) { conv_c32w_debug(enable == 5); return SS_OK; }
  conv_c32_debug(enable);
  return SS_OK;
}
extern "C" int ss_debug_conv_c16(int enable) { conv_c16_debug(enable); return SS_OK; }
extern "C" int ss_debug_rtlin(int grid, int enable) {
  if (grid < 0) return SS_ERR_ARG;
  rtlin_debug(grid, enable);
  return SS_OK;
}
extern "C" int ss_debug_ffn(int grid, int row_tiles_per_wave, int enable) {
  if (grid < 0 || row_tiles_per_wave < 0 || row_tiles_per_wave > 4) return SS_ERR_ARG;
  ffn_fused_debug_grid(grid);
  ffn_fused_debug_rows(row_tiles_per_wave);
  if (enable >= 0) dispatch_edit([enable](Dispatch& d) { d.ffn_fusion = enable ? 1 : 0; });
  return SS_OK;
}

extern "C" int ss_op_layernorm(void* stream, const float* dx, int ldx, float* dy, int ldy, const float* dg,
                               const float* db, int M, int D, float eps) {
  return launch_layernorm(dx, ldx, dy, ldy, dg, db, M, D, eps, (hipStream_t)stream);
}

extern "C" int ss_op_attention(void* stream, const float* dQ, int ldq, const float* dK, int ldk, const float* dV,
                               int ldv, float* dO, int ldo, int Tq, int Tk, int H, float scale, int causal, int chunk,
                               const float* dP, int ldp, const float* du, const float* dv) {
  AttnArgs a;
  a.Q = dQ; a.ldq = ldq; a.K = dK; a.ldk = ldk; a.V = dV; a.ldv = ldv; a.O = dO; a.ldo = ldo;
  a.Tq = Tq; a.Tk = Tk; a.H = H; a.scale = scale; a.causal = causal; a.chunk = chunk;
  a.P = dP; a.ldp = ldp; a.bias_u = du; a.bias_v = dv;
  if (dP) {                                               // test op: one process-wide key-split scratch (callers are serial)
    static void* scratch = nullptr;
    if (!scratch) {
      SS_HIP_CHECK(hipMalloc(&scratch, attention_split_bytes()));
      SS_HIP_CHECK(hipMemset(scratch, 0, attention_split_bytes()));
    }
    attention_bind_split(a, scratch);
  }
  return launch_attention(a, (hipStream_t)stream);
}

extern "C" int ss_debug_attention_split(int v) { attention_debug_split(v); return SS_OK; }

extern "C" int ss_op_dwconv_bn_silu(void* stream, const float* dx, int ldx, float* dy, int ldy, const float* dwt,
                                    int K, const float* mean, const float* var, const float* gamma,
                                    const float* beta, float eps, int T, int C, int chunk) {
  return launch_dwconv_bn_silu(dx, ldx, dy, ldy, dwt, K, mean, var, gamma, beta, eps, T, C, chunk, (hipStream_t)stream);
}

extern "C" int ss_prof_enable(int cls_mask) { prof_enable(cls_mask); return SS_OK; }
extern "C" int ss_prof_reset(void) { prof_reset(); return SS_OK; }
extern "C" int ss_prof_read(int cls, double* ms, double* flops, int64_t* launches, double* bytes) {
  long long n = 0;
  int rc = prof_read(cls, ms, flops, &n, bytes);
  if (launches) *launches = n;
  return rc;
}
extern "C" int ss_prof_totals(int cls, double* flops, double* bytes, int64_t* launches) {
  long long n = 0;
  int rc = prof_totals(cls, flops, bytes, &n);
  if (launches) *launches = n;
  return rc;
}
extern "C" int ss_prof_read_issued(int cls, double* issued_flops) { return prof_read_issued(cls, issued_flops); }
extern "C" int ss_prof_shape_log(int on) { prof_shape_log(on); return SS_OK; }
extern "C" int ss_prof_shape_dump(char* buf, int cap) { return prof_shape_dump(buf, cap); }
extern "C" int ss_prof_num_classes(void) { return kNumTileCfg; }
extern "C" const char* ss_prof_class_name(int cls) { return prof_cfg_name(cls); }

extern "C" int ss_debug_force_tile(int bm, int bn, int ks) {
  // 0 heuristic | 1 first-generation stream-K (bn = 8: XCD groups, ks = grid) | 2 no slab kernel | 3 narrow-stage pairs as two
  // launches | 4 second-generation stream-K (ks = grid) | 5 its split-bf16 form | 6 narrow-stage ResBlocks as separate launches |
  // 32 / 64 / 128 a forced tile of the LDS-tiled kernel (tools/conv_bench.py); anything else is a caller's mistake.
  // (Round 3 had booked BOTH the conv_sk2 hook and the ResBlock A/B on 4, so (4, 0, G) never reached the stream-K launcher.)
  if (!(bm >= 0 && bm <= 6) && bm != 32 && bm != 64 && bm != 128) return SS_ERR_ARG;
  if (bm == 6 || bm == 0) dispatch_edit([bm](Dispatch& d) { d.no_resblock_fusion = (bm == 6); });        // bm = 6: narrow-stage ResBlocks as separate launches (A/B of resblock.hip)
  if (bm == 3 || bm == 0) dispatch_edit([bm](Dispatch& d) { d.no_pair_fusion = (bm == 3); });            // bm = 3: narrow-stage resblock pairs as two launches (A/B of the fused kernel)
  debug_force_tile((bm == 3 || bm == 6) ? 0 : bm, bn, ks);
  return SS_OK;
}
extern "C" int ss_debug_sk_errors(void) { return conv_sk_error_count() + conv_sk2_error_count() + g_mt_timeouts.load(std::memory_order_relaxed); }
