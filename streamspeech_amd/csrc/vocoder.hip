// Unit HiFi-GAN vocoder with duration prediction (reference agent/tts/vocoder.py:48-60, agent/tts/codehifigan.py:56-95,
// fairseq/models/text_to_speech/hifigan.py:52-172): context, generator stack, the single-utterance
// and the ragged-batch forward.
#include "model_internal.hpp"

#include <cstdint>
#include <tuple>


// Transformed weights are a function of the weight blob alone: contexts made over the same blob (HipVocoder.new_context: one per
// concurrent stream) borrow one buffer instead of packing ~16 MB each.  Keyed by (device, blob pointer, signature of
// everything the layout of the transformed buffer depends on: channel plan, ResBlock kernel sizes, the blob's slot offsets), ref-counted:
// a second configuration over a live blob gets its OWN entry (it used to erase the live one and orphan its buffer).
// The blob must stay immutable while any context made over it lives (header): a rewrite in place is not detected.
namespace {
struct WinoShared { DevBuf buf; size_t floats = 0; int refs = 0; };
struct WinoKey {
  int dev; const float* blob; uint64_t sig;
  bool operator<(const WinoKey& o) const { return std::tie(dev, blob, sig) < std::tie(o.dev, o.blob, o.sig); }
};
std::mutex g_wino_mu;
std::map<WinoKey, WinoShared> g_wino;
inline uint64_t fnv(uint64_t h, uint64_t v) { for (int i = 0; i < 8; ++i) { h ^= (v >> (8 * i)) & 0xff; h *= 1099511628211ull; } return h; }
}  // namespace

extern "C" int ss_vocoder_create(const ss_vocoder_config* cfg, const float* d_blob, size_t blob_floats,
                                 const char* const* names, const int64_t* offsets, const int64_t* numels,
                                 int n_slots, ss_vocoder** out) {
  if (!cfg || !d_blob || !out || cfg->n_up > 8 || cfg->n_res > 4) return SS_ERR_ARG;
  ss_vocoder* v = new ss_vocoder();
  v->cfg = *cfg;
  v->sc = new ss_scratch();          // the handle's own scratch set; ss_vocoder_bind_scratch swaps it for a shared one
  int rc = v->wt.build(d_blob, blob_floats, names, offsets, numels, n_slots);
  if (rc != SS_OK) { scratch_unref(v->sc); delete v; return rc; }
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
  if (!w.missing.empty()) { scratch_unref(v->sc); delete v; return SS_ERR_MISSING_WEIGHT; }
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
      if (hipGetDevice(&dev) != hipSuccess) { scratch_unref(v->sc); delete v; return SS_ERR_HIP; }
      std::lock_guard<std::mutex> lk(g_wino_mu);          // (held over the pack: a second context of the same blob waits for it)
      uint64_t sig = fnv(fnv(fnv(14695981039346656037ull, (uint64_t)C0), (uint64_t)cfg->n_up), (uint64_t)cfg->n_res);
      for (int j = 0; j < cfg->n_res; ++j) sig = fnv(sig, (uint64_t)cfg->resblock_kernel_sizes[j]);
      for (size_t i = 0; i < v->rb_c1.size(); ++i) sig = fnv(fnv(sig, (uint64_t)(v->rb_c1[i].w - d_blob)), (uint64_t)(v->rb_c2[i].w - d_blob));
      const WinoKey key{dev, d_blob, sig};
      WinoShared& sh = g_wino[key];
      const bool fresh = sh.refs == 0;
      if (!fresh && sh.floats != need) { scratch_unref(v->sc); delete v; return SS_ERR_ARG; }   // (same signature, another size: cannot happen; the live entry is left alone)
      if (fresh) {
        rc = sh.buf.ensure(need * sizeof(float));
        if (rc != SS_OK) { g_wino.erase(key); scratch_unref(v->sc); delete v; return rc; }
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
        if (fresh) { sh.buf.release(); g_wino.erase(key); }
        scratch_unref(v->sc); delete v; return rc;
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
  if (v->wino_key) {
    std::lock_guard<std::mutex> lk(g_wino_mu);
    for (auto it = g_wino.begin(); it != g_wino.end(); ++it)
      if (it->first.blob == v->wino_key && it->second.buf.f() == v->wino) {
        if (--it->second.refs == 0) { it->second.buf.release(); g_wino.erase(it); }
        break;
      }
  }
  scratch_unref(v->sc);
  delete v;
}

extern "C" int ss_vocoder_bind_scratch(ss_vocoder* v, ss_scratch* sc) {
  if (!v || !sc) return SS_ERR_ARG;
  if (sc != v->sc) {
    sc->refs.fetch_add(1);
    scratch_unref(v->sc);
    v->sc = sc;
  }
  return SS_OK;
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
  SkScope sk_scope(v->sc->skws);
  hipStream_t s = (hipStream_t)stream;
  const ss_vocoder_config& c = v->cfg;
  const int E = c.embedding_dim, Hd = c.dur_hidden;
  // --- embedding + duration predictor (codehifigan.py:56-66, fastspeech2.py:117-151) ---
  RET(v->sc->v_small.ensure(((size_t)K * (E + 2 * Hd + 1) + 2 * (K + 2)) * sizeof(float)));
  float* emb = v->sc->v_small.f();
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
  RET(v->sc->v_ws.ensure((8 * stage_max + (size_t)Fr * E) * sizeof(float)));
  float* frames = v->sc->v_ws.f();
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


extern "C" int ss_batch_vocoder_forward(ss_vocoder* v, void* stream, int B, const int32_t* d_codes, const int32_t* h_K,
                                        int dur_prediction, const int32_t* d_forced_dur, float* d_wav,
                                        int64_t wav_capacity, int32_t* d_dur, int64_t* h_wav_start,
                                        int64_t* h_n_samples) {
  if (!v || B <= 0 || !d_codes || !d_wav || !d_dur) return SS_ERR_ARG;
  SkScope sk_scope(v->sc->skws);
  hipStream_t s = (hipStream_t)stream;
  const ss_vocoder_config& c = v->cfg;
  const int E = c.embedding_dim, Hd = c.dur_hidden;
  for (int b = 0; b < B; ++b)
    if (h_K[b] <= 0) return SS_ERR_ARG;                    // the single-utterance form refuses K = 0 too; callers drop unit-less utterances
  const Offsets ok = prefix(h_K, B);
  const int Kt = ok.total;
  RET(v->sc->v_small.ensure(((size_t)Kt * (E + 2 * Hd + 1) + 2 * (Kt + B + 2)) * sizeof(float)));
  float* emb = v->sc->v_small.f();
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
  RET(v->sc->v_segs.ensure((6 * B + 16 * B) * sizeof(int)));
  int* dk = (int*)v->sc->v_segs.p;
  RET(upload(s, dk, tk));
  RET(launch_gather_rows(d_codes, v->dict, E, emb, Kt, s, v->cfg.num_embeddings));
  const int* forced = d_forced_dur;
  auto sconv = [&](const float* A, int Cin, const ConvW& cw, int Cout, int k, float* Cc, int act) {
    GemmArgs a;
    a.A = A; a.lda = Cin; a.W = cw.w; a.bias = cw.b; a.C = Cc; a.ldc = Cout; a.N = Cout; a.Cin = Cin; a.taps = k;
    a.pad = (k - 1) / 2; a.act = act; a.segs = dk; a.nseg = B; a.max_seg_out = ok.mx; a.M = Kt; a.in_len = Kt;
    a.canon = CANON_SEQ;      // durations are integers (round(exp(.) - 1)): the predictor's convs must not change their summation order with the pack
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
  RET(v->sc->v_ws.ensure((8 * stage_max + (size_t)Ft * E) * sizeof(float)));
  float* frames = v->sc->v_ws.f();
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

