// Ragged-batch twins of the model stages (ss_batch_*): B independent utterances packed along the row axis, no padding anywhere,
// pack-invariant arithmetic (ss_model_set_pack_invariant).  Reference: every stage of agent/speech_to_speech.streamspeech.agent.py
// :425-717 run for B = 1 utterances at a time.
#include "model_internal.hpp"

// =================================================================================================
// Ragged-batch stage twins: B independent utterances packed along the row axis.  No padding exists
// anywhere -- every utterance keeps the B = 1 arithmetic of the single-utterance entry points
// (SURVEY.md H2b); only launches, weight streaming and tile occupancy are shared.
// =================================================================================================
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
  RET(m->sc->seg_buf.ensure(segs.size() * sizeof(int)));
  RET(upload(s, (int*)m->sc->seg_buf.p, segs));
  return launch_fbank_cmvn_batch(d_pcm, pcm_scale, m->fe_window, m->fe_melw, m->fe_mean, m->fe_std, d_feat,
                                 (const int*)m->sc->seg_buf.p, B, mx, s);
}

extern "C" int ss_batch_encoder_forward(ss_model* m, void* stream, int B, const float* d_fbank, const int32_t* h_T,
                                        int attn_chunk, int conv_chunk, float* d_enc_out, int32_t* h_Tp) {
  if (!m || B <= 0) return SS_ERR_ARG;
  SkScope sk_scope(m->sc->skws);
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
  RET(m->sc->seg_buf.ensure(tab.size() * sizeof(int)));
  int* dt = (int*)m->sc->seg_buf.p;
  RET(upload(s, dt, tab));
  const int *d0 = dt, *d1 = dt + 4 * B, *da = dt + 8 * B, *dr = dt + 12 * B;

  const size_t n_h1 = (size_t)M1 * (c.conv_channels / 2), n_x = (size_t)M2 * d, n_f = (size_t)M2 * f,
               n_qkv = (size_t)M2 * 3 * d;
  RET(m->sc->ws.ensure((n_h1 + 3 * n_x + n_f + n_qkv) * sizeof(float)));
  float* h1 = m->sc->ws.f();
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
    const bool fuse_ffn = (canon || (disp().ffn_fusion && M2 >= disp().ffn_min_rows)) && ffn_fused_eligible(d, f, ACT_SILU, M2, d, d, canon) &&
                          e.ffn1_w1.b && e.ffn1_w2.b && e.ffn2_w1.b && e.ffn2_w2.b;
    if (canon && !fuse_ffn) return SS_ERR_ARG;      // never switch FFN forms silently in a pack-invariant context (the two-launch form sums in another order)
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
    at.P = m->pos_proj + (size_t)l * d; at.ldp = Ld; at.p_tmax = c.max_rel_pos; at.bias_u = e.u; at.bias_v = e.v;
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
  SkScope sk_scope(m->sc->skws);
  CanonScope canon_scope(m->pack_invariant ? CANON_SEQ : CANON_NONE);
  hipStream_t s = (hipStream_t)stream;
  const ss_config& c = m->cfg;
  const Offsets o = prefix(h_Tp, B);
  const int V = head == 0 ? c.src_vocab : c.tgt_vocab;
  RET(m->sc->mt_ws.ensure((size_t)o.total * V * sizeof(float)));
  float* logits = m->sc->mt_ws.f();
  std::vector<int> tr(2 * B);
  for (int b = 0; b < B; ++b) { tr[2 * b] = o.off[b]; tr[2 * b + 1] = h_Tp[b]; }
  RET(m->sc->seg_buf.ensure(tr.size() * sizeof(int)));
  RET(upload(s, (int*)m->sc->seg_buf.p, tr));
  RET(linear(s, d_enc_out, c.enc_dim, o.total, head == 0 ? m->ctc_asr : m->ctc_st, V, c.enc_dim, logits, V));
  m->sc->dbg_logits = logits; m->sc->dbg_rows = o.total; m->sc->dbg_cols = V;
  RET(launch_masked_argmax(logits, V, o.total, V, c.pad, c.unk, -1, -1, d_raw, s));
  return launch_ctc_collapse(d_raw, 0, 0, c.pad, d_tokens, d_index, d_counts, s, (const int*)m->sc->seg_buf.p, B);
}

// Batched beam-1 search: all utterances start from [</s>] and advance in lockstep, one row per
// utterance (M = B GEMMs stream every decoder weight once per step for the whole batch).
extern "C" int ss_batch_mt_greedy(ss_model* m, void* stream, int B, const float* d_enc_out, const int32_t* h_Tp,
                                  const int32_t* h_max_len, int min_len, int32_t* h_out_tokens, int out_stride,
                                  int32_t* h_n_out, float* d_feats, int feat_rows) {
  if (!m || B <= 0 || B > 256 || !d_feats) return SS_ERR_ARG;     // (256: the segment tables of the slab kernels)
  SkScope sk_scope(m->sc->skws);
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
  RET(m->sc->mt_cross.ensure((size_t)c.mt_layers * oe.total * 2 * D * sizeof(float)));
  for (int l = 0; l < c.mt_layers; ++l)
    RET(linear(s, d_enc_out, c.enc_dim, oe.total, m->mt[l].cross_kv, 2 * D, c.enc_dim,
               m->sc->mt_cross.f() + (size_t)l * oe.total * 2 * D, 2 * D));
  // caches / scratch
  RET(m->sc->bmt_self.ensure((size_t)c.mt_layers * B * Lcap * 3 * D * sizeof(float)));
  RET(m->sc->mt_ws.ensure(((size_t)B * (3 * D + F + V)) * sizeof(float)));
  float* x = m->sc->mt_ws.f();
  float* h = x + (size_t)B * D;
  float* q2 = h + (size_t)B * D;
  float* ff = q2 + (size_t)B * D;
  float* logits = ff + (size_t)B * F;
  // int tables: tokens [Lcap+1][B], max_len [B], cross segs [B][4], self segs per step [Lcap][B][4]
  const size_t n_tok = (size_t)(Lcap + 1) * B;
  RET(m->sc->seg_buf.ensure((n_tok + B + 4 * B + (size_t)Lcap * 4 * B) * sizeof(int)));
  int* tok = (int*)m->sc->seg_buf.p;
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
      float* cache = m->sc->bmt_self.f() + (size_t)l * B * Lcap * 3 * D;
      float* rows = cache + (size_t)step * 3 * D;                    // row b at + b*Lcap*3D
      AttnArgs at;
      at.Q = rows; at.ldq = Lcap * 3 * D; at.K = cache + D; at.V = cache + 2 * D; at.ldk = at.ldv = 3 * D;
      at.O = h; at.ldo = D; at.H = H; at.scale = 1.f; at.causal = 0;   // cache holds exactly the visible keys
      at.segs = d_self + (size_t)step * 4 * B; at.nseg = B; at.max_q = 1;
      AttnArgs ac;
      ac.Q = q2; ac.ldq = D; ac.K = m->sc->mt_cross.f() + (size_t)l * oe.total * 2 * D; ac.V = ac.K + D; ac.ldk = ac.ldv = 2 * D;
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
  SkScope sk_scope(m->sc->skws);
  CanonScope canon_scope(m->pack_invariant ? CANON_SEQ : CANON_NONE);
  hipStream_t s = (hipStream_t)stream;
  const ss_config& c = m->cfg;
  const int D = c.dec_dim, F = c.dec_ffn, V = c.unit_vocab, H = c.dec_heads, up = c.ctc_upsample;
  for (int b = 0; b < B; ++b)
    if (h_n[b] <= 0) return SS_ERR_ARG;                    // every utterance feeds at least the leading </s> state
  const Offsets on = prefix(h_n, B);
  const int Nn = on.total, U = Nn * up;
  const size_t nx = (size_t)U * D;
  RET(m->sc->ws.ensure((3 * nx + (size_t)U * 3 * D + (size_t)U * F + (size_t)Nn * 2 * D + (size_t)Nn * D +
                    (size_t)U * V + (size_t)U) * sizeof(float)));
  float* x = m->sc->ws.f();
  float* h = x + nx;
  float* q2 = h + nx;
  float* selfbuf = q2 + nx;
  float* ff = selfbuf + (size_t)U * 3 * D;
  float* crosskv = ff + (size_t)U * F;
  float* t2u_out = crosskv + (size_t)Nn * 2 * D;
  float* logits = t2u_out + (size_t)Nn * D;
  int32_t* idx_scratch = reinterpret_cast<int32_t*>(logits + (size_t)U * V);
  // tables: t2u self {off,n,off,n}; unit self {25off,25n,25off,25n}; unit cross {25off,25n,off,n}; rows {25off,25n}
  std::vector<int> tab(14 * B + Nn);            // + the packed row -> row of d_feats map of the gather below
  for (int b = 0; b < B; ++b) {
    const int o = on.off[b], n = h_n[b];
    for (int r = 0; r < n; ++r) tab[14 * B + o + r] = b * feat_rows + r;
    int* a = &tab[4 * b]; a[0] = o; a[1] = n; a[2] = o; a[3] = n;
    int* u = &tab[4 * B + 4 * b]; u[0] = o * up; u[1] = n * up; u[2] = o * up; u[3] = n * up;
    int* x2 = &tab[8 * B + 4 * b]; x2[0] = o * up; x2[1] = n * up; x2[2] = o; x2[3] = n;
    tab[12 * B + 2 * b] = o * up; tab[12 * B + 2 * b + 1] = n * up;
  }
  RET(m->sc->seg_buf.ensure(tab.size() * sizeof(int)));
  int* dt = (int*)m->sc->seg_buf.p;
  RET(upload(s, dt, tab));
  // gather the decoder states of each utterance into packed rows: ONE launch (round 4 issued B device-to-device copies per pack --
  // 2975 __amd_rocclr_copyBuffer launches, 1 % of the one-stream kernel time and 64 more dependent launches per pack)
  for (int b = 0; b < B; ++b)
    if (h_n[b] > feat_rows) return SS_ERR_CAPACITY;
  RET(launch_gather_rows(dt + 14 * B, d_feats, D, x, Nn, s, B * feat_rows));
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
  m->sc->dbg_logits = logits; m->sc->dbg_rows = U; m->sc->dbg_cols = V;
  RET(launch_masked_argmax(logits, V, U, V, c.pad, c.unk, mask_eos ? c.eos : -1, -1, d_raw, s));
  return launch_ctc_collapse(d_raw, 0, V - 1, c.pad, d_tokens, idx_scratch, d_counts, s, dt + 12 * B, B);
}

