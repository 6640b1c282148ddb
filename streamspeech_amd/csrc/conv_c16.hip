// Slab Conv1d for the 16-channel HiFi-GAN stage of a packed batch with the WHOLE weight matrix in registers (C = N = 16 at 320x the
// frame rate; reference fairseq/models/text_to_speech/hifigan.py:52-172) -- the k = 11 ResBlocks of the
// stage, conv by conv, instead of one fused launch per ResBlock (resblock.hip: 0.55-0.56 of the FP32-MFMA peak with its halo
// recompute; conv_c32.hip's measurement: separate convs win where a conv is MFMA-bound, i.e. at k = 11).
//   * a 16 x (16 k) weight matrix is k fragments of 16 B per lane: 11 float4 = 44 registers at k = 11 -- loaded ONCE per workgroup,
//     the contraction then has no weight traffic at all: per tap 4 LDS fragments and 16 MFMAs;
//   * persistent workgroups, three per CU (24 KB of LDS each); per block of 256 output rows the input slab (256 + (k - 1) dil rows
//     x 16 channels, rows padded to 20 floats) is staged once, input leaky-ReLU applied on the way; wave tile 64 rows x 16 columns;
//   * float4 bias / residual / MRF accumulate / mean / output.
// Exact f32, tap-major fmaf chains: differs from resblock.hip / conv_slab.hip by summation order only.
#include "gemm.hpp"

#include <cstdlib>

namespace ss {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

namespace {
constexpr int C1_C = 16;
constexpr int C1_WM = 4;
constexpr int C1_BM = 64 * C1_WM;              // 256 output rows per block
constexpr int C1_LDA = C1_C + 4;
constexpr int C1_MAXHALO = 64;
constexpr int C1_MAXSEG = 256;
[[maybe_unused]] constexpr int C1_NUM_RECORDS = 0x7ffffff0;
[[maybe_unused]] constexpr int C1_NP = ((C1_BM + C1_MAXHALO) * (C1_C / 4) + 255) / 256;   // float4 of a slab per thread (5)
}  // namespace

template <int TAPS, bool LRELU>
__global__ __launch_bounds__(256, 3) void conv_c16_kernel(const GemmArgs p, const int slab_rows) {
#if __HIP_DEVICE_COMPILE__
  constexpr int C = C1_C, BM = C1_BM, LDA = C1_LDA, NP = C1_NP, WM = C1_WM, K = TAPS * C;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sA = smem;                                                        // slab [slab_rows][20]
  int* s_blk = reinterpret_cast<int*>(smem + ((slab_rows * LDA + 3) & ~3));   // block prefix per segment

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int r = lane & 15, g = lane >> 4;

  const int nseg = p.nseg > 0 ? p.nseg : 1;
  if (t == 0) {
    int acc = 0;
    for (int s = 0; s < nseg; ++s) {
      s_blk[s] = acc;
      const int len = p.nseg > 0 ? p.segs[4 * s + 1] : p.M;
      acc += (len + BM - 1) / BM;
    }
    s_blk[nseg] = acc;
  }
  __syncthreads();
  const int nblocks = s_blk[nseg];
  const float slope = p.in_slope;

  int seg = 0, seg_lo = 0, seg_hi = 0, m0 = 0;
  auto locate = [&](int blk) {                 // blocks ascend per workgroup
    while (blk >= s_blk[seg + 1]) ++seg;
    seg_lo = p.nseg > 0 ? p.segs[4 * seg] : 0;
    seg_hi = seg_lo + (p.nseg > 0 ? p.segs[4 * seg + 1] : p.in_len);
    m0 = seg_lo + (blk - s_blk[seg]) * BM;
  };

  int blk = blockIdx.x;
  if (blk >= nblocks) return;
  // the weight matrix: fragment `tap` = 16 B of row r at k = 16 tap + 4 g, for the whole kernel
  f32x4 wf[TAPS];
#pragma unroll
  for (int tap = 0; tap < TAPS; ++tap) wf[tap] = *reinterpret_cast<const f32x4*>(p.W + (size_t)r * K + tap * C + 4 * g);

  for (; blk < nblocks; blk += gridDim.x) {
    locate(blk);
    const int cm0 = m0;
    const int m_hi = p.nseg > 0 ? seg_hi : min(seg_hi, p.M);
    const bool edge = (m0 - p.pad < seg_lo) || (m0 - p.pad + slab_rows > seg_hi);   // zero padding only in an utterance's first / last blocks
    __syncthreads();                                       // previous block's slab reads are done
    // ---- slab: global -> registers (all loads in flight) -> [zero padding, leaky-ReLU] -> LDS.  4 consecutive threads read one
    // 64-B row; branch-free loads from a clamped (always valid) row ----
    {
      f32x4 pre[NP];
#pragma unroll
      for (int u = 0; u < NP; ++u) {
        const int rho = (t >> 2) + 64 * u;
        const int gc = min(max(m0 - p.pad + rho, seg_lo), seg_hi - 1);
        pre[u] = *reinterpret_cast<const f32x4*>(p.A + (size_t)gc * p.lda + (t & 3) * 4);
      }
      float* dst = sA + (t >> 2) * LDA + (t & 3) * 4;
#pragma unroll
      for (int u = 0; u < NP; ++u) {
        const int rho = (t >> 2) + 64 * u;
        f32x4 v = pre[u];
        if (edge) {
          const int gin = m0 - p.pad + rho;
          const bool ok = gin >= seg_lo && gin < seg_hi;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = ok ? v[e] : 0.f;
        }
        if (LRELU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], v[e] * slope);        // 0 < slope < 1 (checked on the host)
        }
        if (rho < slab_rows) *reinterpret_cast<f32x4*>(dst + u * 64 * LDA) = v;
      }
    }
    __syncthreads();

    f32x4 acc[WM];
#pragma unroll
    for (int i = 0; i < WM; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* pa = sA + (wave * 16 * WM + r) * LDA + 4 * g;  // + i*16*LDA + tap*dil*LDA
    const int a_step = p.dil * LDA;
    f32x4 xa[WM];
#pragma unroll
    for (int i = 0; i < WM; ++i) xa[i] = *reinterpret_cast<const f32x4*>(pa + i * 16 * LDA);
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap) {
      f32x4 xb[WM];
      if (tap + 1 < TAPS) {
        pa += a_step;
#pragma unroll
        for (int i = 0; i < WM; ++i) xb[i] = *reinterpret_cast<const f32x4*>(pa + i * 16 * LDA);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < WM; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[tap][e], xa[i][e], acc[i], 0, 0, 0);   // D = W . A^T
      if (tap + 1 < TAPS) {
#pragma unroll
        for (int i = 0; i < WM; ++i) xa[i] = xb[i];
      }
    }

    // ---- epilogue: lane holds 4 consecutive channels (4g .. 4g+3) of row r of row tile i ----
    int le = lane;
    asm volatile("" : "+v"(le));               // addresses derived from `le` cannot be hoisted above the contraction
    const int r_e = le & 15, n = (le >> 4) * 4;
    f32x4 bb = f32x4{0.f, 0.f, 0.f, 0.f};
    if (p.bias) bb = *reinterpret_cast<const f32x4*>(p.bias + n);
    f32x4 rr[WM], rr2[WM];
#pragma unroll
    for (int i = 0; i < WM; ++i) {
      const int mc = min(cm0 + wave * 16 * WM + i * 16 + r_e, m_hi - 1);
      if (p.R) rr[i] = *reinterpret_cast<const f32x4*>(p.R + (size_t)mc * p.ldr + n);
      if (p.R2) rr2[i] = *reinterpret_cast<const f32x4*>(p.R2 + (size_t)mc * p.ldr2 + n);
    }
#pragma unroll
    for (int i = 0; i < WM; ++i) {
      const int m = cm0 + wave * 16 * WM + i * 16 + r_e;
      f32x4 v = acc[i];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] += bb[e];
      if (p.act == ACT_LRELU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * p.act_slope;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] *= p.alpha;
      if (p.R) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += rr[i][e];
      }
      if (p.R2) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = rr2[i][e] + v[e];
      }
      if (p.div > 0.f) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = v[e] / p.div;
      }
      if (m < m_hi) {
        *reinterpret_cast<f32x4*>(p.C + (size_t)m * p.ldc + n) = v;
        if (p.C2) {
          f32x4 w2;
#pragma unroll
          for (int e = 0; e < 4; ++e) w2[e] = v[e] > 0.f ? v[e] : v[e] * p.c2_slope;
          *reinterpret_cast<f32x4*>(p.C2 + (size_t)m * p.ldc2 + n) = w2;
        }
      }
    }
  }
#endif
}

// ---- host side ---------------------------------------------------------------------------------
// (A/B knobs SS_NO_CONV_C16 / SS_CONV_C16_MIN_ROWS and this hook: dispatch.hpp)
void conv_c16_debug(int enable) { if (enable >= 0) dispatch_edit([enable](Dispatch& d) { d.c16_off = enable ? 0 : 1; }); }
bool conv_c16_enabled() { return !disp().c16_off; }

bool conv_c16_eligible(const GemmArgs& a) {
  return !disp().c16_off && a.same_rows && a.stride == 1 && a.chunk == 0 && !a.glu && !a.ln_g && !a.x3 && a.Cin == C1_C && a.N == C1_C &&
         a.lda == C1_C && (a.ldc & 3) == 0 && (!a.R || (a.ldr & 3) == 0) && (!a.R2 || (a.ldr2 & 3) == 0) && (!a.C2 || (a.ldc2 & 3) == 0) &&
         (a.taps == 3 || a.taps == 7 || a.taps == 11) && a.dil >= 1 && (a.taps - 1) * a.dil <= C1_MAXHALO && a.pad >= 0 &&
         a.pad <= (a.taps - 1) * a.dil && a.nseg <= C1_MAXSEG && a.M >= disp().c16_min_rows &&
         slab_rows_ok(a.M) &&
         (a.in_act == ACT_NONE || (a.in_act == ACT_LRELU && a.in_slope > 0.f && a.in_slope < 1.f)) &&
         (a.act == ACT_NONE || a.act == ACT_LRELU);
}

template <int TAPS, bool LRELU>
static int launch_c16_t(const GemmArgs& a, hipStream_t stream) {
  const int slab_rows = C1_BM + (a.taps - 1) * a.dil;
  const size_t lds = (size_t)((slab_rows * C1_LDA + 3) & ~3) * sizeof(float) + (C1_MAXSEG + 2) * sizeof(int);
  SkWorkspace* st = nullptr;                       // (only for the device's CU count, cached per context)
  int rc = sk_workspace_acquire(stream, &st);
  if (rc != SS_OK) return rc;
  const int nseg = a.nseg > 0 ? a.nseg : 1;
  const long long max_blocks = (long long)cdiv(a.M, C1_BM) + nseg;      // upper bound (per-segment round-up)
  static const int occ_env = getenv("SS_CONV_C16_WG_PER_CU") ? atoi(getenv("SS_CONV_C16_WG_PER_CU")) : 0;
  const int occ = occ_env > 0 ? occ_env : 3;       // resident workgroups per CU (<= 168 registers, 25 KB of LDS)
  const int grid = (int)std::min<long long>((long long)occ * st->cus, std::max<long long>(1, max_blocks));
  ProfRec rec{}; bool prof = false;
  rc = prof_begin(a, stream, 26, rec, prof);
  if (rc != SS_OK) return rc;
  hipLaunchKernelGGL((conv_c16_kernel<TAPS, LRELU>), dim3(grid), dim3(256), lds, stream, a, slab_rows);
  SS_LAUNCH_CHECK();
  return prof_end(stream, rec, prof);
}

int launch_conv_c16(const GemmArgs& a, hipStream_t stream) {
  if (!conv_c16_eligible(a)) return SS_ERR_ARG;
  const bool lr = a.in_act == ACT_LRELU;
  if (a.taps == 11) return lr ? launch_c16_t<11, true>(a, stream) : launch_c16_t<11, false>(a, stream);
  if (a.taps == 7) return lr ? launch_c16_t<7, true>(a, stream) : launch_c16_t<7, false>(a, stream);
  return lr ? launch_c16_t<3, true>(a, stream) : launch_c16_t<3, false>(a, stream);
}

}  // namespace ss
