// Slab Conv1d with streamed weights for the 32-channel HiFi-GAN stage of a packed batch (C = N = 32 at 160x the frame rate: the
// stage's 18 ResBlock convs, k = 3 / 7 / 11, dilation 1 / 3 / 5, and the 3-tap polyphase form of the up-conv that leaves it;
// reference fairseq/models/text_to_speech/hifigan.py:52-172) -- conv_c64.hip at half the width.
//
// Round 3 ran this stage as one fused launch per ResBlock (resblock.hip: x read once, result written once, intermediates in LDS)
// at 0.55-0.57 of the FP32-MFMA peak: the fusion pays for its locality with recomputed halo rows (1.17x the MFMA work at k = 11)
// and an 8-wave workgroup whose 16 x 16 units reuse a weight fragment 4-6 times.  With the slab + streamed-weight structure the
// SEPARATE convs are fast enough that the HBM round trips between them are the smaller cost: a conv moves 0.3-0.45 GB (60-90 us)
// and contracts 7 / 16.5 / 26 GFLOP (k = 3 / 7 / 11), so k = 7 and 11 are MFMA-bound without any halo recompute.
//   * persistent workgroups, up to three per CU (44 KB of LDS, <= 168 registers); per block of 256 output rows the input slab
//     (256 + (k - 1) dil rows x 32 channels) is staged ONCE, input leaky-ReLU applied on the way; rows padded to 36 floats;
//   * wave tile 64 rows x 32 columns (8 accumulator tiles); a k-step (one tap, one 16-channel block) is 4 LDS fragments + 2 weight
//     fragments taken from L2 straight into registers for 32 MFMAs; the ring holds the 8 fragments of TWO taps (the tap loop is
//     unrolled by two so that ring slots are static), each slot refilled with the tap two ahead -- after the last taps with the
//     next block's first two;
//   * no barrier, no LDS-DMA piece, no LDS write inside the contraction; float4 bias / residual / MRF accumulate / mean / output.
// Exact f32, tap-major fmaf chains: differs from resblock.hip / conv_slab.hip by summation order only.
#include "gemm.hpp"

#include <cstdlib>
#include <type_traits>

#ifndef C32_FENCE
#define C32_FENCE 1
#endif
#if C32_FENCE
#define C32_STEP_FENCE __builtin_amdgcn_sched_barrier(0)
#else
#define C32_STEP_FENCE do { } while (0)
#endif

namespace ss {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

namespace {
constexpr int C3_C = 32;
constexpr int C3_WM = 4;
constexpr int C3_BM = 64 * C3_WM;              // 256 output rows per block
constexpr int C3_LDA = C3_C + 4;
constexpr int C3_MAXHALO = 64;
constexpr int C3_MAXSEG = 256;
[[maybe_unused]] constexpr int C3_NUM_RECORDS = 0x7ffffff0;
[[maybe_unused]] constexpr int C3_NP = ((C3_BM + C3_MAXHALO) * (C3_C / 4) + 255) / 256;   // float4 of a slab per thread (10)
}  // namespace

template <bool LRELU>
__global__ __launch_bounds__(256, 3) void conv_c32_kernel(const GemmArgs p, const int slab_rows) {
#if __HIP_DEVICE_COMPILE__
  constexpr int C = C3_C, BM = C3_BM, LDA = C3_LDA, NP = C3_NP, WM = C3_WM;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sA = smem;                                                        // slab [slab_rows][36]
  int* s_blk = reinterpret_cast<int*>(smem + ((slab_rows * LDA + 3) & ~3));   // block prefix per segment

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int r = lane & 15, g = lane >> 4;
  const int K = p.taps * C;

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

  // weight fragment f of tap `tap`: channel block cc = f / 2, column tile j = f % 2; lane (r, g) takes 16 B of row 16 j + r
  const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, C3_NUM_RECORDS, 0x00020000);
  const int vo = (r * K + 4 * g) * 4;
  auto wload = [&](int tap, int f) -> f32x4 {
    const int so = __builtin_amdgcn_readfirstlane((((f & 1) * 16) * K + tap * C + (f >> 1) * 16) * 4);
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsW, vo, so, 0);
    return f32x4{__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3])};
  };

  int blk = blockIdx.x;
  if (blk >= nblocks) return;
  const int taps = p.taps;
  f32x4 ring[8];                               // slots 0..3: the even tap in flight, 4..7: the odd one
#pragma unroll
  for (int f = 0; f < 4; ++f) { ring[f] = wload(0, f); ring[4 + f] = wload(taps > 1 ? 1 : 0, f); }

  for (; blk < nblocks; blk += gridDim.x) {
    locate(blk);
    const int cm0 = m0;
    const int m_hi = p.nseg > 0 ? seg_hi : min(seg_hi, p.M);
    const bool edge = (m0 - p.pad < seg_lo) || (m0 - p.pad + slab_rows > seg_hi);   // zero padding only in an utterance's first / last blocks
    __syncthreads();                                       // previous block's slab reads are done
    // ---- slab: global -> registers (all loads in flight) -> [zero padding, leaky-ReLU] -> LDS.  8 consecutive threads read one
    // 128-B row; branch-free loads from a clamped (always valid) row ----
    {
      f32x4 pre[NP];
#pragma unroll
      for (int u = 0; u < NP; ++u) {
        const int rho = (t >> 3) + 32 * u;
        const int gc = min(max(m0 - p.pad + rho, seg_lo), seg_hi - 1);
        pre[u] = *reinterpret_cast<const f32x4*>(p.A + (size_t)gc * p.lda + (t & 7) * 4);
      }
      float* dst = sA + (t >> 3) * LDA + (t & 7) * 4;
#pragma unroll
      for (int u = 0; u < NP; ++u) {
        const int rho = (t >> 3) + 32 * u;
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
        if (rho < slab_rows) *reinterpret_cast<f32x4*>(dst + u * 32 * LDA) = v;
      }
    }
    __syncthreads();

    f32x4 acc[WM][2];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const float* pa = sA + (wave * 16 * WM + r) * LDA + 4 * g;  // + i*16*LDA + tap*dil*LDA + cc*16
    const int a_step = p.dil * LDA;
    f32x4 xa[WM];
#pragma unroll
    for (int i = 0; i < WM; ++i) xa[i] = *reinterpret_cast<const f32x4*>(pa + i * 16 * LDA);
    // one tap out of ring slots [4 S, 4 S + 4): two k-steps of 32 MFMAs; the slot is refilled with the tap two ahead (beyond the
    // last tap: with tap S of the next block, whose slot parity it is)
    auto do_tap = [&](int tap, auto slot) {
      constexpr int S = decltype(slot)::value;
      const int tap_fill = tap + 2 < taps ? tap + 2 : (taps > 1 ? S : 0);
      const float* pa_next = tap + 1 < taps ? pa + a_step : pa;     // (after the last tap: a harmless re-read)
#pragma unroll
      for (int cc = 0; cc < 2; ++cc) {
        f32x4 xb[WM];
#pragma unroll
        for (int i = 0; i < WM; ++i)
          xb[i] = *reinterpret_cast<const f32x4*>((cc < 1 ? pa : pa_next) + i * 16 * LDA + (cc < 1 ? 16 : 0));
        f32x4 wf[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int f = cc * 2 + j;
          wf[j] = ring[4 * S + f];
          ring[4 * S + f] = wload(tap_fill, f);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[j][e], xa[i][e], acc[i][j], 0, 0, 0);   // D = W . A^T
#pragma unroll
        for (int i = 0; i < WM; ++i) xa[i] = xb[i];
        C32_STEP_FENCE;
      }
      pa = pa_next;
    };
    int tap = 0;
#pragma unroll 1
    for (; tap + 1 < taps; tap += 2) {
      do_tap(tap, std::integral_constant<int, 0>{});
      do_tap(tap + 1, std::integral_constant<int, 1>{});
    }
    if (tap < taps) do_tap(tap, std::integral_constant<int, 0>{});

    // ---- epilogue: lane holds 4 consecutive channels (4g .. 4g+3 of column tile j) of row r of row tile i ----
    int le = lane;
    asm volatile("" : "+v"(le));               // addresses derived from `le` cannot be hoisted above the contraction
    const int r_e = le & 15, g_e = le >> 4;
    f32x4 bb[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      bb[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (p.bias) bb[j] = *reinterpret_cast<const f32x4*>(p.bias + j * 16 + g_e * 4);
    }
    f32x4 rr[WM][2], rr2[WM][2];
#pragma unroll
    for (int i = 0; i < WM; ++i) {
      const int mc = min(cm0 + wave * 16 * WM + i * 16 + r_e, m_hi - 1);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if (p.R) rr[i][j] = *reinterpret_cast<const f32x4*>(p.R + (size_t)mc * p.ldr + j * 16 + g_e * 4);
        if (p.R2) rr2[i][j] = *reinterpret_cast<const f32x4*>(p.R2 + (size_t)mc * p.ldr2 + j * 16 + g_e * 4);
      }
    }
#pragma unroll
    for (int i = 0; i < WM; ++i) {
      const int m = cm0 + wave * 16 * WM + i * 16 + r_e;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int n = j * 16 + g_e * 4;
        f32x4 v = acc[i][j];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += bb[j][e];
        if (p.act == ACT_LRELU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * p.act_slope;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] *= p.alpha;
        if (p.R) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += rr[i][j][e];
        }
        if (p.R2) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = rr2[i][j][e] + v[e];
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
  }
#endif
}

// ---- host side ---------------------------------------------------------------------------------
// (A/B knobs SS_NO_CONV_C32 / SS_CONV_C32_MIN_ROWS and this hook: dispatch.hpp)
void conv_c32_debug(int enable) { if (enable >= 0) dispatch_edit([enable](Dispatch& d) { d.c32_off = enable ? 0 : 1; }); }
bool conv_c32_enabled() { return !disp().c32_off; }

bool conv_c32_eligible(const GemmArgs& a) {
  return !disp().c32_off && a.same_rows && a.stride == 1 && a.chunk == 0 && !a.glu && !a.ln_g && !a.x3 && a.Cin == C3_C && a.N == C3_C &&
         a.lda == C3_C && (a.ldc & 3) == 0 && (!a.R || (a.ldr & 3) == 0) && (!a.R2 || (a.ldr2 & 3) == 0) && (!a.C2 || (a.ldc2 & 3) == 0) &&
         a.taps >= 1 && a.dil >= 1 && (a.taps - 1) * a.dil <= C3_MAXHALO && a.pad >= 0 && a.pad <= (a.taps - 1) * a.dil &&
         a.nseg <= C3_MAXSEG && a.M >= disp().c32_min_rows && slab_rows_ok(a.M) &&
         (a.in_act == ACT_NONE || (a.in_act == ACT_LRELU && a.in_slope > 0.f && a.in_slope < 1.f)) &&
         (a.act == ACT_NONE || a.act == ACT_LRELU);
}

template <bool LRELU>
static int launch_c32_t(const GemmArgs& a, hipStream_t stream) {
  const int slab_rows = C3_BM + (a.taps - 1) * a.dil;
  const size_t lds = (size_t)((slab_rows * C3_LDA + 3) & ~3) * sizeof(float) + (C3_MAXSEG + 2) * sizeof(int);
  SkWorkspace* st = nullptr;                       // (only for the device's CU count, cached per context)
  int rc = sk_workspace_acquire(stream, &st);
  if (rc != SS_OK) return rc;
  const int nseg = a.nseg > 0 ? a.nseg : 1;
  const long long max_blocks = (long long)cdiv(a.M, C3_BM) + nseg;      // upper bound (per-segment round-up)
  static const int occ_env = getenv("SS_CONV_C32_WG_PER_CU") ? atoi(getenv("SS_CONV_C32_WG_PER_CU")) : 0;
  const int occ = occ_env > 0 ? occ_env : (int)std::min<size_t>(3, (158 * 1024) / lds);   // resident workgroups per CU
  const int grid = (int)std::min<long long>((long long)occ * st->cus, std::max<long long>(1, max_blocks));
  ProfRec rec{}; bool prof = false;
  rc = prof_begin(a, stream, 25, rec, prof);
  if (rc != SS_OK) return rc;
  hipLaunchKernelGGL((conv_c32_kernel<LRELU>), dim3(grid), dim3(256), lds, stream, a, slab_rows);
  SS_LAUNCH_CHECK();
  return prof_end(stream, rec, prof);
}

int launch_conv_c32(const GemmArgs& a, hipStream_t stream) {
  if (!conv_c32_eligible(a)) return SS_ERR_ARG;
  return a.in_act == ACT_LRELU ? launch_c32_t<true>(a, stream) : launch_c32_t<false>(a, stream);
}

}  // namespace ss
