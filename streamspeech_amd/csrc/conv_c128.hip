// Slab Conv1d with streamed weights for the 128-channel HiFi-GAN stage of a packed batch (C = N = 128 at 20x the frame rate: the
// stage's 18 ResBlock convs, k = 3 / 7 / 11, dilation 1 / 3 / 5; reference fairseq/models/text_to_speech/hifigan.py:52-172,
// SURVEY.md §8a row a15).  The 64-channel form (conv_c64.hip) carried over to twice the width:
//   * one persistent workgroup per CU, four waves (one per SIMD, up to 512 registers); per block of 192 output rows the input slab
//     (192 + (k - 1) dil rows x 128 channels = up to 128 KB of LDS, rows padded to 132 floats) is staged ONCE with the input
//     leaky-ReLU applied on the way, so the producer writes no pre-activated twin tensor; the NEXT block's slab is requested into
//     registers (31 float4 per thread) before the contraction and lands under it;
//   * wave tile 48 rows x 128 columns (24 accumulator tiles); a k-step (one tap, one 16-channel block) is 3 LDS fragments + 8
//     weight fragments taken from L2 straight into registers (ring of 16 = two k-steps = 6144 MFMA cycles ahead, wrapping from
//     the last tap to the first) for 96 MFMAs; no barrier, no LDS-DMA piece, no LDS write inside the contraction;
//   * the first tap is peeled out of the tap loop: vmcnt counts in order and a loop has one wait immediate at its top, so with
//     the slab prefetch issued in front of the loop hipcc would wait for the whole HBM round trip at the loop's first ring use; in
//     the peeled straight-line tap the counts are exact and the prefetch stays in flight (conv_c64.hip's measurement history).
// Exact f32, tap-major fmaf chains: differs from the stream-K path by summation order only.
#include "gemm.hpp"

#include <cstdlib>

#ifndef C128_FENCE
#define C128_FENCE 1
#endif
#if C128_FENCE
#define C128_STEP_FENCE __builtin_amdgcn_sched_barrier(0)
#else
#define C128_STEP_FENCE do { } while (0)
#endif

namespace ss {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

namespace {
constexpr int C8_C = 128;
constexpr int C8_WM = 3;                       // 16-row MFMA tiles per wave
constexpr int C8_BM = 64 * C8_WM;              // 192 output rows per block
constexpr int C8_LDA = C8_C + 4;
constexpr int C8_MAXHALO = 50;                 // (taps - 1) * dil (k = 11, dil = 5)
constexpr int C8_MAXSEG = 256;
[[maybe_unused]] constexpr int C8_RING = 16;
[[maybe_unused]] constexpr int C8_NUM_RECORDS = 0x7ffffff0;
[[maybe_unused]] constexpr int C8_NP = ((C8_BM + C8_MAXHALO) * (C8_C / 4) + 255) / 256;   // float4 of a slab per thread (31)
}  // namespace

template <bool LRELU>
__global__ __launch_bounds__(256, 1) void conv_c128_kernel(const GemmArgs p, const int slab_rows) {
#if __HIP_DEVICE_COMPILE__
  constexpr int C = C8_C, BM = C8_BM, LDA = C8_LDA, NP = C8_NP, WM = C8_WM, NT = C / 16;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sA = smem;                                                        // slab [slab_rows][132]
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

  // weight fragment f of tap `tap`: channel block cc = f / 8, column tile j = f % 8; lane (r, g) takes 16 B of row 16 j + r
  const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, C8_NUM_RECORDS, 0x00020000);
  const int vo = (r * K + 4 * g) * 4;
  auto wload = [&](int tap, int f) -> f32x4 {
    const int so = __builtin_amdgcn_readfirstlane((((f & 7) * 16) * K + tap * C + (f >> 3) * 16) * 4);
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsW, vo, so, 0);
    return f32x4{__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3])};
  };

  int blk = blockIdx.x;
  if (blk >= nblocks) return;
  f32x4 ring[C8_RING];
#pragma unroll
  for (int f = 0; f < C8_RING; ++f) ring[f] = wload(0, f);

  // slab of a block: global -> registers.  32 consecutive threads read one 512-B row (thread t: row t / 32 + 8 u, chunk t % 32);
  // branch-free loads from a clamped (always valid) row, zero padding applied when the registers are staged
  f32x4 pre[NP];
  int pm0 = 0, plo = 0, phi = 0;
  auto slab_request = [&](int b2) {
    locate(b2);
    pm0 = m0; plo = seg_lo; phi = seg_hi;
#pragma unroll
    for (int u = 0; u < NP; ++u) {
      const int rho = (t >> 5) + 8 * u;
      const int gc = min(max(m0 - p.pad + rho, seg_lo), seg_hi - 1);
      pre[u] = *reinterpret_cast<const f32x4*>(p.A + (size_t)gc * p.lda + (t & 31) * 4);
    }
  };
  slab_request(blk);

  for (; blk < nblocks; blk += gridDim.x) {
    const int cm0 = pm0;
    const int m_hi = p.nseg > 0 ? phi : min(phi, p.M);
    const bool edge = (pm0 - p.pad < plo) || (pm0 - p.pad + slab_rows > phi);   // zero padding only in an utterance's first / last blocks
    __syncthreads();                                       // previous block's slab reads are done
    {
      float* dst = sA + (t >> 5) * LDA + (t & 31) * 4;
#pragma unroll
      for (int u = 0; u < NP; ++u) {
        const int rho = (t >> 5) + 8 * u;
        f32x4 v = pre[u];
        if (edge) {
          const int gin = pm0 - p.pad + rho;
          const bool ok = gin >= plo && gin < phi;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = ok ? v[e] : 0.f;
        }
        if (LRELU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], v[e] * slope);        // 0 < slope < 1 (checked on the host)
        }
        if (rho < slab_rows) *reinterpret_cast<f32x4*>(dst + u * 8 * LDA) = v;
      }
    }
    __syncthreads();
    if (blk + (int)gridDim.x < nblocks) slab_request(blk + gridDim.x);       // lands under this block's contraction

    f32x4 acc[WM][NT];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const float* pa = sA + (wave * 16 * WM + r) * LDA + 4 * g;  // + i*16*LDA + tap*dil*LDA + cc*16
    const int a_step = p.dil * LDA;
    f32x4 xa[WM];
#pragma unroll
    for (int i = 0; i < WM; ++i) xa[i] = *reinterpret_cast<const f32x4*>(pa + i * 16 * LDA);
    auto do_tap = [&](int tap) {
      const int tap_next = tap + 1 < p.taps ? tap + 1 : 0;          // after the last tap: the next block's first fragments
      const float* pa_next = tap + 1 < p.taps ? pa + a_step : pa;   // (after the last tap: a harmless re-read)
#pragma unroll
      for (int cc = 0; cc < 8; ++cc) {
        f32x4 xb[WM];
#pragma unroll
        for (int i = 0; i < WM; ++i)
          xb[i] = *reinterpret_cast<const f32x4*>((cc < 7 ? pa : pa_next) + i * 16 * LDA + (cc < 7 ? (cc + 1) * 16 : 0));
        f32x4 wf[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const int f = cc * 8 + j;                                  // 64 fragments per tap
          wf[j] = ring[f % C8_RING];
          ring[f % C8_RING] = wload(f + C8_RING < 64 ? tap : tap_next, (f + C8_RING) & 63);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[j][e], xa[i][e], acc[i][j], 0, 0, 0);   // D = W . A^T
#pragma unroll
        for (int i = 0; i < WM; ++i) xa[i] = xb[i];
        C128_STEP_FENCE;
      }
      pa = pa_next;
    };
    do_tap(0);
#pragma unroll 1
    for (int tap = 1; tap < p.taps; ++tap) do_tap(tap);

    // ---- epilogue: lane holds 4 consecutive channels (4g .. 4g+3 of column tile j) of row r of row tile i ----
    int le = lane;
    asm volatile("" : "+v"(le));               // addresses derived from `le` cannot be hoisted above the contraction
    const int r_e = le & 15, g_e = le >> 4;
#pragma unroll
    for (int i = 0; i < WM; ++i) {
      const int m = cm0 + wave * 16 * WM + i * 16 + r_e;
      const int mc = min(m, m_hi - 1);
      f32x4 rr[NT], rr2[NT];
      if (p.R) {
#pragma unroll
        for (int j = 0; j < NT; ++j) rr[j] = *reinterpret_cast<const f32x4*>(p.R + (size_t)mc * p.ldr + j * 16 + g_e * 4);
      }
      if (p.R2) {
#pragma unroll
        for (int j = 0; j < NT; ++j) rr2[j] = *reinterpret_cast<const f32x4*>(p.R2 + (size_t)mc * p.ldr2 + j * 16 + g_e * 4);
      }
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int n = j * 16 + g_e * 4;
        f32x4 v = acc[i][j];
        if (p.bias) {
          const f32x4 b = *reinterpret_cast<const f32x4*>(p.bias + n);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += b[e];
        }
        if (p.act == ACT_LRELU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * p.act_slope;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] *= p.alpha;
        if (p.R) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += rr[j][e];
        }
        if (p.R2) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = rr2[j][e] + v[e];
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
static int g_c128_off = getenv("SS_NO_CONV_C128") && atoi(getenv("SS_NO_CONV_C128")) ? 1 : 0;   // A/B knob: the C = 128 stage on conv_sk2<128> as in round 3
static long long g_c128_min_rows = getenv("SS_CONV_C128_MIN_ROWS") ? atoll(getenv("SS_CONV_C128_MIN_ROWS")) : 24576;   // >= half a block per CU
void conv_c128_debug(int enable) { if (enable >= 0) g_c128_off = enable ? 0 : 1; }
bool conv_c128_enabled() { return !g_c128_off; }

bool conv_c128_eligible(const GemmArgs& a) {
  return !g_c128_off && a.same_rows && a.stride == 1 && a.chunk == 0 && !a.glu && !a.ln_g && !a.x3 && a.Cin == C8_C && a.N == C8_C &&
         a.lda == C8_C && (a.ldc & 3) == 0 && (!a.R || (a.ldr & 3) == 0) && (!a.R2 || (a.ldr2 & 3) == 0) && (!a.C2 || (a.ldc2 & 3) == 0) &&
         a.taps >= 1 && a.dil >= 1 && (a.taps - 1) * a.dil <= C8_MAXHALO && a.pad >= 0 && a.pad <= (a.taps - 1) * a.dil &&
         a.nseg <= C8_MAXSEG && a.M >= g_c128_min_rows && ((size_t)(a.M + a.pad + 512) * a.lda) * 4 < 0x7ff00000ull &&
         (size_t)a.taps * C8_C * C8_C * 4 < 0x7ff00000ull && (a.in_act == ACT_NONE || (a.in_act == ACT_LRELU && a.in_slope > 0.f && a.in_slope < 1.f)) &&
         (a.act == ACT_NONE || a.act == ACT_LRELU);
}

template <bool LRELU>
static int launch_c128_t(const GemmArgs& a, hipStream_t stream) {
  const int slab_rows = C8_BM + (a.taps - 1) * a.dil;
  const size_t lds = (size_t)((slab_rows * C8_LDA + 3) & ~3) * sizeof(float) + (C8_MAXSEG + 2) * sizeof(int);
  SS_MAX_LDS_ONCE((&conv_c128_kernel<LRELU>), 132 * 1024);
  SkWorkspace* st = nullptr;                       // (only for the device's CU count, cached per context)
  int rc = sk_workspace_acquire(stream, &st);
  if (rc != SS_OK) return rc;
  const int nseg = a.nseg > 0 ? a.nseg : 1;
  const long long max_blocks = (long long)cdiv(a.M, C8_BM) + nseg;      // upper bound (per-segment round-up)
  const int grid = (int)std::min<long long>((long long)st->cus, std::max<long long>(1, max_blocks));   // one workgroup per CU
  ProfRec rec{}; bool prof = false;
  rc = prof_begin(a, stream, 25, rec, prof);
  if (rc != SS_OK) return rc;
  hipLaunchKernelGGL((conv_c128_kernel<LRELU>), dim3(grid), dim3(256), lds, stream, a, slab_rows);
  SS_LAUNCH_CHECK();
  return prof_end(stream, rec, prof);
}

int launch_conv_c128(const GemmArgs& a, hipStream_t stream) {
  if (!conv_c128_eligible(a)) return SS_ERR_ARG;
  return a.in_act == ACT_LRELU ? launch_c128_t<true>(a, stream) : launch_c128_t<false>(a, stream);
}

}  // namespace ss
