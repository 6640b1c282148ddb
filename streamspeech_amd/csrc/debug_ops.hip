// Op-level unit-test entry points (ss_op_*), test / A-B hooks (ss_debug_*) and the profiler's C ABI (ss_prof_*).
#include "model_internal.hpp"

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

extern "C" int ss_debug_last_logits(ss_model* m, void* stream, float* d_out, int64_t cap_floats, int* h_rows, int* h_cols) {
  if (!m || !h_rows || !h_cols) return SS_ERR_ARG;
  *h_rows = m->sc->dbg_rows; *h_cols = m->sc->dbg_cols;
  if (!d_out) return SS_OK;                                   // size query
  if (!m->sc->dbg_logits || cap_floats < (int64_t)m->sc->dbg_rows * m->sc->dbg_cols) return SS_ERR_CAPACITY;
  SS_HIP_CHECK(hipMemcpyAsync(d_out, m->sc->dbg_logits, (size_t)m->sc->dbg_rows * m->sc->dbg_cols * sizeof(float), hipMemcpyDeviceToDevice,
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
extern "C" int64_t ss_debug_enc_step_launches(void) { return (int64_t)enc_step_launch_count(); }
extern "C" int ss_debug_rtlin(int grid, int enable) {
  if (grid < 0) return SS_ERR_ARG;
  rtlin_debug(grid, enable);
  return SS_OK;
}
extern "C" int ss_debug_ffn(int grid, int row_tiles_per_wave, int enable) {
  if (grid < 0 || row_tiles_per_wave < -1 || row_tiles_per_wave > 4) return SS_ERR_ARG;     // row tiles: 1..4 force, 0 process default, -1 keep
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
extern "C" int ss_debug_attention_q16(int v) { attention_debug_q16(v); return SS_OK; }

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

