// Context creation / weight binding, front-end and the SINGLE-UTTERANCE stages of the StreamSpeech S2ST path
// (include/streamspeech_hip.h).  Host code only queues kernels on the caller's stream; the only device->host syncs are the
// ones documented per entry point.  Ragged-batch twins: batch.hip; vocoder: vocoder.hip; test entry points: debug_ops.hip.
#include "model_internal.hpp"

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

// The projected rel-pos table is a function of the weight blob alone: handles made over the same blob (one per language AND stream
// since scratch sets are separate objects) borrow one buffer instead of 67 MB each.  Keyed by (device, table slot, projection slot,
// shape), ref-counted -- the same scheme as the vocoder's Winograd weights (vocoder.hip).
namespace {
struct PosShared { DevBuf buf; int refs = 0; };
struct PosKey {
  int dev; const float* table; const float* w; int rows, cols;
  bool operator<(const PosKey& o) const { return std::tie(dev, table, w, rows, cols) < std::tie(o.dev, o.table, o.w, o.rows, o.cols); }
};
std::mutex g_pos_mu;
std::map<PosKey, PosShared> g_pos;
}  // namespace

extern "C" int ss_abi_version(void) { return SS_ABI_VERSION; }

extern "C" const char* ss_error_string(int code) {
  switch (code) {
    case SS_OK: return "ok";
    case SS_ERR_HIP: return "HIP runtime error";
    case SS_ERR_ARG: return "invalid argument";
    case SS_ERR_MISSING_WEIGHT: return "weight slot missing or wrong size";
    case SS_ERR_CAPACITY: return "output capacity too small";
    case SS_ERR_SCRATCH_CAP: return "scratch set would grow past its cap (ss_scratch_set_cap)";
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
  m->sc = new ss_scratch();            // the handle's own scratch set; ss_model_bind_scratch swaps it for a shared one
  SkScope sk_scope(m->sc->skws);       // the pos_proj GEMM below may take a stream-K kernel
  int rc = m->wt.build(d_blob, blob_floats, names, offsets, numels, n_slots);
  if (rc != SS_OK) { ss_model_destroy(m); return rc; }
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
  if (!w.missing.empty()) { ss_model_destroy(m); return SS_ERR_MISSING_WEIGHT; }

  // projected rel-pos table for every layer at once: [2Tm-1, d] x [L*d, d]^T (linear_pos has no
  // bias, espnet_multihead_attention.py:125).  Depends only on the relative offset, so it is
  // computed once here and sliced per utterance.
  const int rows = 2 * Tm - 1, Ld = cfg->enc_layers * d;
  {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { ss_model_destroy(m); return SS_ERR_HIP; }
    std::lock_guard<std::mutex> lk(g_pos_mu);        // (held over the GEMM: a second handle over the same blob waits for it)
    const PosKey key{dev, m->pos_table, m->pos_w, rows, Ld};
    PosShared& sh = g_pos[key];
    if (sh.refs == 0) {
      rc = sh.buf.ensure((size_t)rows * Ld * sizeof(float));
      if (rc == SS_OK) {
        Lin lp{m->pos_w, nullptr};
        rc = linear(nullptr, m->pos_table, d, rows, lp, Ld, d, sh.buf.f(), Ld);
      }
      if (rc == SS_OK && hipDeviceSynchronize() != hipSuccess) rc = SS_ERR_HIP;
      if (rc != SS_OK) { sh.buf.release(); g_pos.erase(key); }
    }
    if (rc == SS_OK) { ++sh.refs; m->pos_proj = sh.buf.f(); m->pos_key_dev = dev; }
  }
  if (rc == SS_OK) rc = scratch_fit_model(m->sc, *cfg);
  if (rc != SS_OK) { ss_model_destroy(m); return rc; }
  *out = m;
  return SS_OK;
}

extern "C" void ss_model_destroy(ss_model* m) {
  if (!m) return;
  if (m->pos_proj) {
    std::lock_guard<std::mutex> lk(g_pos_mu);
    for (auto it = g_pos.begin(); it != g_pos.end(); ++it)
      if (it->first.dev == m->pos_key_dev && it->second.buf.f() == m->pos_proj) {
        if (--it->second.refs == 0) { it->second.buf.release(); g_pos.erase(it); }
        break;
      }
  }
  scratch_unref(m->sc);
  delete m;
}

// ---- scratch sets (see ss_scratch in model_internal.hpp) -------------------------------------------------------------------------
extern "C" int ss_scratch_create(ss_scratch** out) {
  if (!out) return SS_ERR_ARG;
  *out = new ss_scratch();
  return SS_OK;
}
extern "C" void ss_scratch_destroy(ss_scratch* sc) { scratch_unref(sc); }
extern "C" int ss_scratch_set_cap(ss_scratch* sc, size_t max_bytes) {
  if (!sc) return SS_ERR_ARG;
  sc->acct.cap = max_bytes;
  return SS_OK;
}
extern "C" size_t ss_scratch_bytes(ss_scratch* sc) { return sc ? sc->acct.used : 0; }
extern "C" int ss_scratch_trim(ss_scratch* sc, size_t keep_bytes) {
  if (!sc) return SS_ERR_ARG;
  SS_HIP_CHECK(hipDeviceSynchronize());              // nothing queued may still read what is let go
  std::vector<DevBuf*> bufs = sc->trimmable();
  std::sort(bufs.begin(), bufs.end(), [](const DevBuf* a, const DevBuf* b) { return a->bytes > b->bytes; });
  for (DevBuf* b : bufs) {
    if (sc->acct.used <= keep_bytes) break;
    b->release();
  }
  // state that lived in the released buffers: a stateful sequence (ss_mt_begin ... ss_mt_append, ss_encoder_stream_*) starts over
  sc->mt_Tp = 0; sc->mt_len = 0; sc->mt_enc = nullptr;
  sc->es_cap = 0; sc->es_final = 0; sc->es_achunk = sc->es_cchunk = -1;
  sc->dbg_logits = nullptr; sc->dbg_rows = sc->dbg_cols = 0;
  return SS_OK;
}
extern "C" int ss_model_bind_scratch(ss_model* m, ss_scratch* sc) {
  if (!m || !sc) return SS_ERR_ARG;
  int rc = scratch_fit_model(sc, m->cfg);
  if (rc != SS_OK) return rc;
  if (sc != m->sc) {
    sc->refs.fetch_add(1);
    scratch_unref(m->sc);
    m->sc = sc;
  }
  return SS_OK;
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

extern "C" int ss_encoder_out_len(int T) {
  const int t1 = conv_out_len(T, 5, 2);
  return conv_out_len(t1, 5, 2);
}

extern "C" int ss_encoder_forward(ss_model* m, void* stream, const float* d_fbank, int T, int attn_chunk,
                                  int conv_chunk, float* d_enc_out) {
  if (!m || T <= 0) return SS_ERR_ARG;
  SkScope sk_scope(m->sc->skws);
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
  RET(m->sc->ws.ensure((n_h1 + 3 * n_x + n_f + n_qkv) * sizeof(float)));
  float* h1 = m->sc->ws.f();
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
  const float* P = m->pos_proj + (size_t)(c.max_rel_pos - T2) * Ld;

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
  m->sc->es_final = 0; m->sc->es_achunk = -1; m->sc->es_cchunk = -1;
  m->sc->es_final_prev = 0;      // (a deferred check that is still outstanding must not bring the old utterance's rows back)
  return SS_OK;
}

extern "C" int ss_encoder_stream_set_tail(ss_model* m, int unsettled_fbank_frames) {
  if (!m || unsettled_fbank_frames < 0) return SS_ERR_ARG;
  m->sc->es_tail = unsettled_fbank_frames;
  return SS_OK;
}

// The time-out check of the persistent layer launches: waits for the stream, reads the pinned error word; on a time-out the scratch
// set leaves the persistent form, the rows of the failed call stop being final, and the caller must repeat the call.
static int es_check(ss_model* m, hipStream_t s, bool* repeat) {
  ss_scratch* sc = m->sc;
  *repeat = false;
  SS_HIP_CHECK(hipStreamSynchronize(s));
  sc->es_pending = 0;
  unsigned e = sc->es_err_host ? *static_cast<volatile unsigned*>(sc->es_err_host) : 0u;   // written by a timed-out wait only (system scope; visible after the synchronisation)
  const unsigned real = e;                                   // an injected time-out (test hook) is not counted among the process's time-outs
  if (sc->es_inject) { e += 1u; sc->es_inject = 0; }
  if (!e) return SS_OK;
  fprintf(stderr, "streamspeech_hip: persistent encoder-layer launch timed out (its %d workgroups were not all resident); "
                  "this context falls back to one launch per op\n", ES_G);
  g_mt_timeouts.fetch_add((int)real, std::memory_order_relaxed);
  if (sc->es_err_host) *sc->es_err_host = 0u;
  if (sc->es_step.p) {
    unsigned* es_sync = reinterpret_cast<unsigned*>(sc->es_step.f() + (size_t)ES_G * ES_MAXR * ES_D + (size_t)ES_MAXR * ES_D);
    SS_HIP_CHECK(hipMemsetAsync(es_sync, 0, 512, s));
  }
  sc->es_bar = 0;
  sc->es_step_off = 1;
  sc->es_final = sc->es_final_prev;
  *repeat = true;
  return SS_OK;
}

extern "C" int ss_encoder_stream_set_deferred(ss_model* m, int on) {
  if (!m) return SS_ERR_ARG;
  m->sc->es_deferred = on ? 1 : 0;
  return SS_OK;
}

extern "C" int ss_encoder_stream_status(ss_model* m, void* stream, int32_t* repeat) {
  if (!m || !repeat) return SS_ERR_ARG;
  *repeat = 0;
  if (!m->sc->es_pending && !m->sc->es_inject) return SS_OK;
  bool rep = false;
  RET(es_check(m, (hipStream_t)stream, &rep));
  *repeat = rep ? 1 : 0;
  return SS_OK;
}

extern "C" int ss_debug_enc_step_inject_timeout(ss_model* m) {
  if (!m) return SS_ERR_ARG;
  m->sc->es_inject = 1;
  return SS_OK;
}

extern "C" int ss_encoder_stream_forward(ss_model* m, void* stream, const float* d_fbank, int T, int attn_chunk,
                                         int conv_chunk, float* d_enc_out, int32_t* n_final, int32_t* n_computed) {
  if (!m || T <= 0) return SS_ERR_ARG;
  SkScope sk_scope(m->sc->skws);
  hipStream_t s = (hipStream_t)stream;
  if (m->sc->es_pending) {        // the caller skipped ss_encoder_stream_status: settle the previous call first (its rows stop being final if it failed)
    bool rep = false;
    RET(es_check(m, s, &rep));
  }
  const ss_config& c = m->cfg;
  const int d = c.enc_dim, f = c.enc_ffn, k = c.conv_kernel, Ld = c.enc_layers * d, L = c.enc_layers;
  const int T1 = conv_out_len(T, k, 2), T2 = conv_out_len(T1, k, 2);
  if (T2 <= 0 || T2 > c.max_rel_pos) return SS_ERR_CAPACITY;
  const int cchunk = (conv_chunk > 0 && conv_chunk < 999) ? conv_chunk : 0;
  const int achunk_cfg = (attn_chunk > 0 && attn_chunk < 999999) ? attn_chunk : 0;   // as configured (not clipped by T2)
  const int achunk = (achunk_cfg > 0 && achunk_cfg < T2) ? achunk_cfg : 0;
  if (m->sc->es_achunk != achunk_cfg || m->sc->es_cchunk != cchunk) { m->sc->es_final = 0; m->sc->es_achunk = achunk_cfg; m->sc->es_cchunk = cchunk; }
  if (m->sc->es_final > T2) m->sc->es_final = 0;           // audio got shorter: a new utterance without reset
  if (m->sc->es_cap < T2) {                             // grow (contents are only needed below es_final: keep them)
    const int cap = std::min(c.max_rel_pos, std::max(2 * T2, 256));
    DevBuf nq, ng, no;
    RET(nq.ensure((size_t)L * cap * 3 * d * sizeof(float)));
    RET(ng.ensure((size_t)L * cap * d * sizeof(float)));
    RET(no.ensure((size_t)cap * d * sizeof(float)));
    if (m->sc->es_final > 0) {
      for (int l = 0; l < L; ++l) {
        SS_HIP_CHECK(hipMemcpyAsync(nq.f() + (size_t)l * cap * 3 * d, m->sc->es_qkv.f() + (size_t)l * m->sc->es_cap * 3 * d,
                                    (size_t)m->sc->es_final * 3 * d * sizeof(float), hipMemcpyDeviceToDevice, s));
        SS_HIP_CHECK(hipMemcpyAsync(ng.f() + (size_t)l * cap * d, m->sc->es_glu.f() + (size_t)l * m->sc->es_cap * d,
                                    (size_t)m->sc->es_final * d * sizeof(float), hipMemcpyDeviceToDevice, s));
      }
      SS_HIP_CHECK(hipMemcpyAsync(no.f(), m->sc->es_out.f(), (size_t)m->sc->es_final * d * sizeof(float), hipMemcpyDeviceToDevice, s));
      SS_HIP_CHECK(hipStreamSynchronize(s));
    }
    m->sc->es_qkv.release(); m->sc->es_glu.release(); m->sc->es_out.release();
    m->sc->es_qkv = nq; m->sc->es_glu = ng; m->sc->es_out = no;
    nq.p = nullptr; ng.p = nullptr; no.p = nullptr;
    m->sc->es_cap = cap;
  }
  const int cap = m->sc->es_cap;
  const int r0 = m->sc->es_final;                       // first row to (re)compute
  const int n = T2 - r0;
  if (n_computed) *n_computed = n;

  const size_t n_h1 = (size_t)T1 * (c.conv_channels / 2);
  const size_t n_x = (size_t)T2 * d, n_f = (size_t)n * f;
  RET(m->sc->ws.ensure((n_h1 + 3 * n_x + n_f) * sizeof(float)));
  float* h1 = m->sc->ws.f();
  float* g0 = h1 + n_h1;                // subsampler output [T2, d]
  float* h = g0 + n_x;                  // LN output / attention context (tail rows, indexed from 0)
  float* g2 = h + n_x;                  // depthwise output, absolute rows
  float* ff = g2 + n_x;                 // FFN hidden (tail rows)
  float* x = d_enc_out + (size_t)r0 * d;  // running activations of the tail rows live in the output buffer

  if (n > 0) {
    // Only rows >= r0 of the subsampler output are read below: conv 2 starts at row r0, conv 1 at the first row conv 2 can reach
    // (2 r0 - pad).  Both keep absolute row indices (GemmArgs::m_begin), so halo rows and the chunk rule are those of the full call;
    // with the fbank rows cached by the front-end, nothing in front of the layers grows with the length of the utterance any more.
    GemmArgs a;
    a.A = d_fbank; a.lda = c.input_feat; a.W = m->sub0.w; a.bias = m->sub0.b; a.C = h1; a.ldc = c.conv_channels / 2;
    a.M = T1; a.N = c.conv_channels; a.Cin = c.input_feat; a.taps = k; a.stride = 2; a.pad = k / 2;
    a.in_len = T; a.chunk = cchunk; a.glu = 1;
    a.m_begin = std::min(std::max(0, 2 * r0 - k / 2), T1 - 1);
    RET(launch_conv_gemm(a, s));
    GemmArgs b;
    b.A = h1; b.lda = c.conv_channels / 2; b.W = m->sub1.w; b.bias = m->sub1.b; b.C = g0; b.ldc = d;
    b.M = T2; b.N = 2 * d; b.Cin = c.conv_channels / 2; b.taps = k; b.stride = 2; b.pad = k / 2;
    b.in_len = T1; b.chunk = cchunk; b.glu = 1;
    b.m_begin = r0;
    RET(launch_conv_gemm(b, s));
  }
  if (r0 > 0)
    SS_HIP_CHECK(hipMemcpyAsync(d_enc_out, m->sc->es_out.f(), (size_t)r0 * d * sizeof(float), hipMemcpyDeviceToDevice, s));
  // The layers as two persistent launches around the attention kernel (enc_step.hip) when this scratch set runs the persistent forms
  // (ss_mt_set_persistent: a context that decodes one utterance at a time -- the agents) and the call has at most 48 rows to compute;
  // a bounded wait that times out is counted, the set falls back to one launch per op for good and THIS call is repeated that way.
  bool es_persistent = n > 0 && n <= ES_MAXR && m->sc->mt_persistent > 0 && !m->sc->es_step_off && !g_no_enc_step && d == ES_D && f == ES_F &&
                       c.enc_heads * 64 == d && c.dw_kernel <= 31 && (size_t)T2 * 3 * d * 4 < 0x7ff00000ull;
  if (es_persistent && !m->sc->es_step.p) {
    RET(m->sc->es_step.ensure(enc_step_scratch_bytes()));
    SS_HIP_CHECK(hipMemsetAsync(m->sc->es_step.p, 0, enc_step_scratch_bytes(), s));
    m->sc->es_bar = 0;
  }
  if (es_persistent && !m->sc->es_err_host) {
    SS_HIP_CHECK(hipHostMalloc((void**)&m->sc->es_err_host, 64));
    *m->sc->es_err_host = 0u;
  }
  float* es_part = es_persistent ? m->sc->es_step.f() : nullptr;
  float* es_g2 = es_persistent ? es_part + (size_t)ES_G * ES_MAXR * ES_D : nullptr;
  unsigned* es_sync = es_persistent ? reinterpret_cast<unsigned*>(es_g2 + (size_t)ES_MAXR * ES_D) : nullptr;      // [64] flags, then the error word
  if (n > 0) {
    RET(linear(s, g0 + (size_t)r0 * d, d, n, m->enc_linear, d, d, x, d));
    const float* P = m->pos_proj + (size_t)(c.max_rel_pos - T2) * Ld;
    for (int l = 0; l < L; ++l) {
      const EncLayer& e = m->enc[l];
      float* qkv = m->sc->es_qkv.f() + (size_t)l * cap * 3 * d;      // absolute rows
      float* glu = m->sc->es_glu.f() + (size_t)l * cap * d;
      if (es_persistent) {
        EsArgs a;
        a.w = EsLayerW{e.ffn1_ln.g, e.ffn1_ln.b, e.ffn1_w1.w, e.ffn1_w1.b, e.ffn1_w2.w, e.ffn1_w2.b,
                       e.attn_ln.g, e.attn_ln.b, e.qkv.w, e.qkv.b, e.out.w, e.out.b,
                       e.conv_ln.g, e.conv_ln.b, e.pw1.w, e.dw_wt, e.bn_mean, e.bn_var, e.bn_g, e.bn_b, e.pw2.w,
                       e.ffn2_ln.g, e.ffn2_ln.b, e.ffn2_w1.w, e.ffn2_w1.b, e.ffn2_w2.w, e.ffn2_w2.b, e.final_ln.g, e.final_ln.b};
        a.x = x; a.qkv = qkv; a.glu = glu; a.hctx = h; a.part = es_part; a.g2 = es_g2; a.bar = es_sync;
        a.err = es_sync + 64; a.err_host = m->sc->es_err_host; a.n = n; a.r0 = r0; a.T2 = T2; a.cchunk = cchunk; a.dwk = c.dw_kernel;
        a.ph0 = 0; a.ph1 = 2; a.bar_base = m->sc->es_bar;
        RET(launch_enc_step(a, s));
        m->sc->es_bar += (unsigned)ES_G * 2u;
        AttnArgs at;
        at.Q = qkv + (size_t)r0 * 3 * d; at.K = qkv + d; at.V = qkv + 2 * d; at.ldq = at.ldk = at.ldv = 3 * d;
        at.O = h; at.ldo = d; at.Tq = n; at.Tk = T2; at.q0 = r0; at.H = c.enc_heads; at.scale = 0.125f;
        at.chunk = achunk; at.P = P + (size_t)l * d; at.ldp = Ld; at.bias_u = e.u; at.bias_v = e.v;
        RET(bind_attn_split(m, at, s));
        RET(launch_attention(at, s));
        a.ph0 = 4; a.ph1 = 9; a.bar_base = m->sc->es_bar;
        RET(launch_enc_step(a, s));
        m->sc->es_bar += (unsigned)ES_G * 5u;
        continue;
      }
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
  if (es_persistent) {
    m->sc->es_final_prev = r0;
    if (m->sc->es_deferred) {
      m->sc->es_pending = 1;      // ss_encoder_stream_status (or the next forward) settles it
    } else {
      bool rep = false;
      RET(es_check(m, s, &rep));
      if (rep)                    // a workgroup of some launch was not resident: this call again, one launch per op
        return ss_encoder_stream_forward(m, stream, d_fbank, T, attn_chunk, conv_chunk, d_enc_out, n_final, n_computed);
    }
  }
  const int nf = std::max(r0, stream_final_rows(T, T1, T2, k, achunk_cfg, cchunk, c.dw_kernel, m->sc->es_tail));
  if (nf > r0)
    SS_HIP_CHECK(hipMemcpyAsync(m->sc->es_out.f() + (size_t)r0 * d, d_enc_out + (size_t)r0 * d, (size_t)(nf - r0) * d * sizeof(float),
                                hipMemcpyDeviceToDevice, s));
  m->sc->es_final = nf;
  if (n_final) *n_final = nf;
  return SS_OK;
}

// ---- CTC heads ---------------------------------------------------------------------------------
extern "C" int ss_ctc_greedy(ss_model* m, void* stream, int head, const float* d_enc_out, int Tp,
                             int32_t* d_raw, int32_t* d_tokens, int32_t* d_index, int32_t* d_count,
                             float* d_logits) {
  if (!m || Tp <= 0 || head < 0 || head > 1) return SS_ERR_ARG;
  SkScope sk_scope(m->sc->skws);
  hipStream_t s = (hipStream_t)stream;
  const ss_config& c = m->cfg;
  const int V = head == 0 ? c.src_vocab : c.tgt_vocab;
  float* logits = d_logits;
  if (!logits) {
    RET(m->sc->mt_ws.ensure((size_t)Tp * V * sizeof(float)));
    logits = m->sc->mt_ws.f();
  }
  RET(linear(s, d_enc_out, c.enc_dim, Tp, head == 0 ? m->ctc_asr : m->ctc_st, V, c.enc_dim, logits, V));
  RET(launch_masked_argmax(logits, V, Tp, V, c.pad, c.unk, -1, -1, d_raw, s));
  return launch_ctc_collapse(d_raw, Tp, 0, c.pad, d_tokens, d_index, d_count, s);
}

extern "C" int ss_mt_begin(ss_model* m, void* stream, const float* d_enc_out, int Tp) {
  if (!m || Tp <= 0) return SS_ERR_ARG;
  SkScope sk_scope(m->sc->skws);
  hipStream_t s = (hipStream_t)stream;
  const ss_config& c = m->cfg;
  const int D = c.dec_dim;
  RET(m->sc->mt_cross.ensure((size_t)c.mt_layers * Tp * 2 * D * sizeof(float)));
  for (int l = 0; l < c.mt_layers; ++l)
    RET(linear(s, d_enc_out, c.enc_dim, Tp, m->mt[l].cross_kv, 2 * D, c.enc_dim,
               m->sc->mt_cross.f() + (size_t)l * Tp * 2 * D, 2 * D));
  m->sc->mt_Tp = Tp;
  m->sc->mt_len = 0;
  m->sc->mt_enc = d_enc_out;
  return SS_OK;
}

// Bounded-wait time-outs of persistent MT decode steps, process-wide (reported by ss_debug_sk_errors next to the stream-K
// counters).  The step has its OWN device error word per context (it used to alias the stream-K time-out counter,
// so one MT time-out made every later stream-K launch of that workspace bail out); the host collects and clears it whenever
// the persistent form is switched (which both fall-back paths do).
std::atomic<int> g_mt_timeouts{0};
static unsigned* mt_err_word(ss_model* m) {
  return reinterpret_cast<unsigned*>(static_cast<char*>(m->sc->mt_gran.p) + mt_step_granule_bytes() + 16);
}
static int mt_collect_errors(ss_model* m) {
  if (!m->sc->mt_gran.p) return SS_OK;
  // on the stream the persistent step was last launched on (torch streams are non-blocking: the legacy null stream orders
  // nothing against them -- ADVICE r4): the read sees every earlier launch's error word, the clear lands before any later one
  hipStream_t s = m->sc->mt_last_stream;
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
  if (m->sc->mt_persistent > 0) RET(mt_collect_errors(m));
  m->sc->mt_persistent = workgroups;
  if (workgroups > 0) m->sc->es_step_off = 0;      // asking for the persistent forms again also re-arms the persistent layer launches
  return SS_OK;
}

extern "C" int ss_mt_get_persistent(ss_model* m) { return m ? m->sc->mt_persistent : SS_ERR_ARG; }

extern "C" int ss_debug_mt_inject_timeout(ss_model* m) {
  if (!m) return SS_ERR_ARG;
  m->sc->mt_inject_timeout = 1;
  return SS_OK;
}

extern "C" int ss_mt_truncate(ss_model* m, int len) {
  if (!m || len < 0 || len > m->sc->mt_len) return SS_ERR_ARG;
  m->sc->mt_len = len;
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
    w.selfbuf = m->sc->mt_self.f() + (size_t)l * c.max_tgt_pos * 3 * D;
    w.cross = m->sc->mt_cross.f() + (size_t)l * m->sc->mt_Tp * 2 * D;
  }
  a.lnf_g = m->mt_ln.g; a.lnf_b = m->mt_ln.b; a.emb = m->mt_emb; a.pos_table = m->mt_pos;
  a.feats = scratch_feats;
  a.gran = reinterpret_cast<mt_u64*>(m->sc->mt_gran.p); a.err = mt_err_word(m);
  a.Tp = m->sc->mt_Tp; a.V = c.tgt_vocab; a.pad = c.pad; a.eos = c.eos;
  a.emb_scale = sqrtf((float)D);
  return SS_OK;
}
// n consecutive epochs for a launch of n steps (0 is what a fresh granule holds: skipped)
static unsigned mt_next_epochs(ss_model* m, int n) {
  if (m->sc->mt_epoch + (unsigned)n + 1u < m->sc->mt_epoch || m->sc->mt_epoch == 0u) m->sc->mt_epoch = 0u;      // wrap: start over above 0
  const unsigned first = m->sc->mt_epoch + 1u;
  m->sc->mt_epoch += (unsigned)n;
  return first;
}
static int mt_inject(ss_model* m, MtStepArgs& a, hipStream_t s) {      // test hook: this one launch sees a time-out that already happened
  if (!m->sc->mt_inject_timeout) return SS_OK;
  m->sc->mt_inject_timeout = 0;
  a.err = reinterpret_cast<unsigned*>(static_cast<char*>(m->sc->mt_gran.p) + mt_step_granule_bytes());
  SS_HIP_CHECK(hipMemsetAsync(a.err, 0x01, sizeof(unsigned), s));
  return SS_OK;
}
static bool mt_persistent_ok(const ss_model* m) {
  const ss_config& c = m->cfg;
  return m->sc->mt_persistent > 0 && c.mt_layers == MT_L && c.dec_dim == MT_D && c.dec_ffn == MT_F && c.dec_heads == MT_H;
}
static int mt_gran_ensure(ss_model* m, hipStream_t s) {
  if (!m->sc->mt_gran.p) {
    RET(m->sc->mt_gran.ensure(mt_step_granule_bytes() + 64));   // + one spare error word (ss_debug_mt_inject_timeout) + the step's own
    SS_HIP_CHECK(hipMemsetAsync(m->sc->mt_gran.p, 0, mt_step_granule_bytes() + 64, s));
    SS_HIP_CHECK(hipStreamSynchronize(s));
  }
  return SS_OK;
}

extern "C" int ss_mt_append(ss_model* m, void* stream, const int32_t* d_tokens, int n, int pos0, int ban_eos,
                            int force_eos, float* d_feats, int32_t* d_next, int n_tail_pad) {
  if (!m || n <= 0 || pos0 < 0 || pos0 > m->sc->mt_len || m->sc->mt_Tp <= 0) return SS_ERR_ARG;
  SkScope sk_scope(m->sc->skws);
  const ss_config& c = m->cfg;
  if (pos0 + n + 2 > c.max_tgt_pos) return SS_ERR_CAPACITY;
  hipStream_t s = (hipStream_t)stream;
  const int D = c.dec_dim, F = c.dec_ffn, V = c.tgt_vocab;
  RET(m->sc->mt_ws.ensure(((size_t)n * (4 * D + F) + V) * sizeof(float)));
  float* x = m->sc->mt_ws.f();
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
    m->sc->mt_last_stream = s;
    RET(launch_mt_step(a, m->sc->mt_persistent, s));
    m->sc->mt_len = pos0 + n;
    return SS_OK;
  }
  // sqrt(D) * E[tok] + sinusoid(position), positions start at padding_idx + 1 (transformer_decoder.py:297-326)
  RET(launch_embed_tokens(d_tokens, m->mt_emb, m->mt_pos, sqrtf((float)D), pos0 + c.pad + 1, x, n, D, s, 1, c.pad, V));
  for (int l = 0; l < c.mt_layers; ++l) {
    float* selfbuf = m->sc->mt_self.f() + (size_t)l * c.max_tgt_pos * 3 * D;
    const float* cross = m->sc->mt_cross.f() + (size_t)l * m->sc->mt_Tp * 2 * D;
    RET(dec_layer(s, c, m->mt[l], x, n, pos0, selfbuf, true, cross, m->sc->mt_Tp, h, q2, ff, n_tail_pad, 0));
  }
  float* fo = d_feats ? d_feats : feats;
  m->sc->mt_len = pos0 + n;
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
  SkScope sk_scope(m->sc->skws);
  const ss_config& c = m->cfg;
  if (max_len + 3 > c.max_tgt_pos) return SS_ERR_CAPACITY;
  for (int i = 0; i < n_prefix; ++i)
    if (h_prefix[i] < 0 || h_prefix[i] >= c.tgt_vocab) return SS_ERR_ARG;   // nn.Embedding's IndexError in the reference
  hipStream_t s = (hipStream_t)stream;
  constexpr int kCheck = 4;
  const int D = c.dec_dim;
  RET(ss_mt_begin(m, stream, d_enc_out, Tp));
  int32_t* tok = reinterpret_cast<int32_t*>(m->sc->mt_tok.p);
  int32_t* host = m->sc->mt_tok_host;
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
      a.pos0 = first; a.n_steps = n_steps; a.min_len = min_len; a.max_len = max_len; a.search = 1;
      RET(mt_inject(m, a, s));
      a.epoch = mt_next_epochs(m, n_steps);
      m->sc->mt_last_stream = s;
      RET(launch_mt_step(a, m->sc->mt_persistent, s));
    }
    const int lo = start + 1, hi = max_len + 1;             // generated tokens live at chain indices lo .. hi
    SS_HIP_CHECK(hipMemcpyAsync(host + lo, tok + lo, (size_t)(hi - lo + 1) * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    SS_HIP_CHECK(hipStreamSynchronize(s));
    int eos_at = -1;
    for (int i = lo; i <= hi && eos_at < 0; ++i) {
      if (host[i] < 0) {                                    // a bounded wait of the persistent kernel timed out
        fprintf(stderr, "streamspeech_hip: persistent MT decode step timed out (its %d workgroups were not all resident); "
                        "this context falls back to one launch per op\n", m->sc->mt_persistent);
        RET(mt_collect_errors(m));
        m->sc->mt_persistent = 0;
        return ss_mt_greedy(m, stream, d_enc_out, Tp, h_prefix, n_prefix, max_len, min_len, h_out_tokens, h_n_out, d_feats, h_n_feats);
      }
      if (host[i] == c.eos) eos_at = i;
    }
    const int end = eos_at >= 0 ? eos_at : hi;              // (position max_len forces </s>: eos_at is always found)
    const int n_out = end - start;
    for (int i = 0; i < n_out; ++i) h_out_tokens[i] = host[start + 1 + i];
    *h_n_out = n_out;
    if (h_n_feats) *h_n_feats = end;
    m->sc->mt_len = end;
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
        if (host[i] < 0 && m->sc->mt_persistent > 0) {          // mt_step.hip: a bounded wait of the persistent step timed out
          fprintf(stderr, "streamspeech_hip: persistent MT decode step timed out (its %d workgroups were not all resident); "
                          "this context falls back to one launch per op\n", m->sc->mt_persistent);
          RET(mt_collect_errors(m));
          m->sc->mt_persistent = 0;
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
  m->sc->mt_len = end;
  return SS_OK;
}

// ---- T2U encoder + CTC unit decoder ------------------------------------------------------------
extern "C" int ss_t2u_units(ss_model* m, void* stream, const float* d_mt_feats, int n, int t2u_causal,
                            int mask_eos, int32_t* d_raw, int32_t* d_tokens, int32_t* d_count, float* d_logits,
                            int n_tail_pad) {
  if (!m || n <= 0) return SS_ERR_ARG;
  SkScope sk_scope(m->sc->skws);
  hipStream_t s = (hipStream_t)stream;
  const ss_config& c = m->cfg;
  const int D = c.dec_dim, F = c.dec_ffn, U = n * c.ctc_upsample, V = c.unit_vocab;
  const size_t nx = (size_t)U * D;
  // t2u: x,h,self(3D) on n rows; unit: x,h,q2 on U rows, self 3D on U rows, ff U*F, cross kv n*2D, logits U*V
  const size_t total = 3 * nx + (size_t)U * 3 * D + (size_t)U * F + (size_t)n * 2 * D + (size_t)n * D +
                       (d_logits ? 0 : (size_t)U * V) + (size_t)U;
  RET(m->sc->ws.ensure(total * sizeof(float)));
  float* x = m->sc->ws.f();
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

extern "C" int ss_model_set_pack_invariant(ss_model* m, int on) {
  if (!m) return SS_ERR_ARG;
  m->pack_invariant = on ? 1 : 0;
  return SS_OK;
}
extern "C" int ss_model_get_pack_invariant(ss_model* m) { return m ? m->pack_invariant : SS_ERR_ARG; }

