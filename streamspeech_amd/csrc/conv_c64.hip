// Slab Conv1d with streamed weights for the 64-channel HiFi-GAN stage of a packed batch (C = N = 64 at 80x the frame rate:
// the stage's 18 ResBlock convs, k = 3 / 7 / 11, dilation 1 / 3 / 5, and the 3-tap polyphase form of the up-conv that leaves it;
// reference fairseq/models/text_to_speech/hifigan.py:52-172).
//
// Until round 4 these launches ran on conv_sk2<64> -- 30 % of the dominant class's time at 0.50 / 0.62 / 0.66 of the FP32-MFMA
// peak (k = 3 / 7 / 11) against 0.73 for the 128-wide variant: with N = 64 every A row chunk staged by LDS-DMA feeds only four
// column tiles, so the per-k-step cost of the staging (8 of the 10 DMA pieces are A rows, restaged once per TAP), the counted wait
// and the barrier weigh twice as much, and the k = 3 convs are HBM-bound on six tensor passes per ResBlock pair (the pre-activated
// twin outputs included).  This kernel combines the narrow stages' slab idea (conv_slab.hip) with the weight streaming of
// ffn.hip / rtlin.hip:
//   * persistent workgroups, TWO per CU; per block of 256 (192 when
//     two 256-row slabs do not fit a CU's LDS) output rows the input slab
//     (block + (k - 1) dil rows x 64 channels) goes global -> registers -> LDS ONCE, the input leaky-ReLU applied on the way (once
//     per element -- no pre-activated twin tensor has to be written by the producer or read here); every tap reads the same slab
//     at a row offset (rows padded to 68 floats: conflict-free ds_read_b128 fragments, addresses linear in the tap).
//     Measured alternatives (profiles/r04_c64_bench*.txt): ONE workgroup per CU with the next slab and the residual prefetched
//     into registers across the contraction (512 registers per wave, first tap peeled so that hipcc's in-order vmcnt waits do
//     not serialise the prefetch) ran the contraction at the MFMA issue rate (31.5 us per tap at 576k rows = 150 TF/s) but paid
//     ~11 us per block of un-overlapped staging / epilogue / store drain: k = 3 195 us against 173 for two in-phase workgroups
//     and 177-181 on stream-K;
//   * the weight matrix (up to 180 KB at k = 11: no room in LDS) is never staged: each wave takes its W fragments from L2 straight
//     into registers -- one buffer load of 16 B per lane with the (tap, channel block, column tile) part of the address in the
//     wave-uniform soffset, a ring of 8 fragments = two k-steps (4096 MFMA cycles) ahead, wrapping from the last tap to the first
//     so that the next block starts with its fragments in flight; all four waves of a workgroup (and the co-resident workgroup)
//     read the same fragments, so L2 serves each once per CU;
//   * no barrier, no LDS-DMA piece and no LDS write inside the contraction: a k-step is 4 LDS + 4 L2 fragments for 64 MFMAs;
//   * wave tile 64 rows x 64 columns (16 accumulator tiles), swapped MFMA operands (D = W . A^T) -> float4 bias / residual / MRF
//     accumulate / output along the channels.
// Exact f32; per element the fmaf chain runs tap-major (conv_sk2: channel-block-major), i.e. results differ from the stream-K path
// by summation order only.
#include "gemm.hpp"

#include <cstdlib>

#ifndef C64_FENCE
#define C64_FENCE 1      // see ffn.hip: without a fence per step hipcc sinks every weight load to just before its use
#endif
#if C64_FENCE
#define C64_STEP_FENCE __builtin_amdgcn_sched_barrier(0)
#else
#define C64_STEP_FENCE do { } while (0)
#endif

namespace ss {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

namespace {
constexpr int C6_C = 64;                       // input channels = output channels
constexpr int C6_BM_MAX = 256;                 // output rows per block: 64 WM (WM = 16-row MFMA tiles per wave: 4 | 3)
constexpr int C6_LDA = C6_C + 4;               // padded slab row (floats)
constexpr int C6_MAXHALO = 64;                 // (taps - 1) * dil <= 64 (k = 11, dil = 5: 50)
constexpr int C6_MAXSEG = 256;
[[maybe_unused]] constexpr int C6_RING = 8;             // weight fragments in flight per wave: two k-steps (4096 MFMA cycles) ahead
[[maybe_unused]] constexpr int C6_NUM_RECORDS = 0x7ffffff0;
[[maybe_unused]] constexpr int C6_NP = ((C6_BM_MAX + C6_MAXHALO) * (C6_C / 4) + 255) / 256;   // float4 of a slab per thread (<= 20)
}  // namespace

template <bool LRELU, int WM>
__global__ __launch_bounds__(256, 2) void conv_c64_kernel(const GemmArgs p, const int slab_rows) {
#if __HIP_DEVICE_COMPILE__
  constexpr int C = C6_C, BM = 64 * WM, LDA = C6_LDA, NP = C6_NP;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sA = smem;                                                        // slab [slab_rows][68]
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
    m0 = seg_lo + (blk - s_blk[seg]) * BM;     // first output row (packed coordinates)
  };

  // ---- weight fragments: L2 -> registers.  Fragment f of tap `tap`: channel block cc = f / 4, column tile j = f % 4 ----
  const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, C6_NUM_RECORDS, 0x00020000);
  const int vo = (r * K + 4 * g) * 4;
  auto wload = [&](int tap, int f) -> f32x4 {
    const int so = __builtin_amdgcn_readfirstlane((((f & 3) * 16) * K + tap * C + (f >> 2) * 16) * 4);
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsW, vo, so, 0);
    return f32x4{__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3])};
  };

  int blk = blockIdx.x;
  if (blk >= nblocks) return;
  // (Two co-resident workgroups that start together run in phase; starting the second half of the grid half a contraction late
  // was measured and changed nothing -- profiles/r04_c64_bench_stagger.txt vs _nostagger.txt -- so there is no stagger.)
  f32x4 ring[C6_RING];
#pragma unroll
  for (int f = 0; f < C6_RING; ++f) ring[f] = wload(0, f);

  for (; blk < nblocks; blk += gridDim.x) {
    locate(blk);
    const int cm0 = m0;
    const int m_hi = p.nseg > 0 ? seg_hi : min(seg_hi, p.M);
    // rows of the slab outside the utterance (= the conv's zero padding) exist only in its first and last blocks
    const bool edge = (m0 - p.pad < seg_lo) || (m0 - p.pad + slab_rows > seg_hi);
    __syncthreads();                                       // previous block's slab reads are done
    // ---- slab: global -> registers (all loads in flight) -> [zero padding, leaky-ReLU] -> LDS.  16 consecutive threads read one
    // 256-B row; branch-free loads from a clamped (always valid) row ----
    {
      f32x4 pre[NP];
#pragma unroll
      for (int u = 0; u < NP; ++u) {
        const int rho = (t >> 4) + 16 * u;
        const int gc = min(max(m0 - p.pad + rho, seg_lo), seg_hi - 1);
        pre[u] = *reinterpret_cast<const f32x4*>(p.A + (size_t)gc * p.lda + (t & 15) * 4);
      }
      float* dst = sA + (t >> 4) * LDA + (t & 15) * 4;
#pragma unroll
      for (int u = 0; u < NP; ++u) {
        const int rho = (t >> 4) + 16 * u;
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
        if (rho < slab_rows) *reinterpret_cast<f32x4*>(dst + u * 16 * LDA) = v;
      }
    }
    __syncthreads();

    f32x4 acc[WM][4];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const float* pa = sA + (wave * 16 * WM + r) * LDA + 4 * g;  // + i*16*LDA + tap*dil*LDA + cc*16
    const int a_step = p.dil * LDA;
    f32x4 xa[WM];
#pragma unroll
    for (int i = 0; i < WM; ++i) xa[i] = *reinterpret_cast<const f32x4*>(pa + i * 16 * LDA);
#pragma unroll 1
    for (int tap = 0; tap < p.taps; ++tap) {
      const int tap_next = tap + 1 < p.taps ? tap + 1 : 0;          // after the last tap: the next block's first fragments
      const float* pa_next = tap + 1 < p.taps ? pa + a_step : pa;   // (after the last tap: a harmless re-read)
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) {
        f32x4 xb[WM];
#pragma unroll
        for (int i = 0; i < WM; ++i)
          xb[i] = *reinterpret_cast<const f32x4*>((cc < 3 ? pa : pa_next) + i * 16 * LDA + (cc < 3 ? (cc + 1) * 16 : 0));
        f32x4 wf[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int f = cc * 4 + j;
          wf[j] = ring[f % C6_RING];
          ring[f % C6_RING] = wload(f + C6_RING < 16 ? tap : tap_next, (f + C6_RING) & 15);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[j][e], xa[i][e], acc[i][j], 0, 0, 0);   // D = W . A^T
#pragma unroll
        for (int i = 0; i < WM; ++i) xa[i] = xb[i];
        C64_STEP_FENCE;
      }
      pa = pa_next;
    }

    // ---- epilogue: lane holds 4 consecutive channels (4g .. 4g+3 of column tile j) of row r of row tile i; a row tile's residual
    // operands are all requested before the first is used ----
    int le = lane;
    asm volatile("" : "+v"(le));               // addresses derived from `le` cannot be hoisted above the contraction
    const int r_e = le & 15, g_e = le >> 4;
    f32x4 bb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      bb[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (p.bias) bb[j] = *reinterpret_cast<const f32x4*>(p.bias + j * 16 + g_e * 4);
    }
#pragma unroll
    for (int i = 0; i < WM; ++i) {
      const int m = cm0 + wave * 16 * WM + i * 16 + r_e;
      const int mc = min(m, m_hi - 1);
      f32x4 rr[4], rr2[4];
      if (p.R) {
#pragma unroll
        for (int j = 0; j < 4; ++j) rr[j] = *reinterpret_cast<const f32x4*>(p.R + (size_t)mc * p.ldr + j * 16 + g_e * 4);
      }
      if (p.R2) {
#pragma unroll
        for (int j = 0; j < 4; ++j) rr2[j] = *reinterpret_cast<const f32x4*>(p.R2 + (size_t)mc * p.ldr2 + j * 16 + g_e * 4);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
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
// (A/B knobs SS_NO_CONV_C64 / SS_CONV_C64_MIN_ROWS and this hook: dispatch.hpp)
void conv_c64_debug(int enable) { if (enable >= 0) dispatch_edit([enable](Dispatch& d) { d.c64_off = enable ? 0 : 1; }); }
bool conv_c64_enabled() { return !disp().c64_off; }

bool conv_c64_eligible(const GemmArgs& a) {
  return !disp().c64_off && a.same_rows && a.stride == 1 && a.chunk == 0 && !a.glu && !a.ln_g && !a.x3 && a.Cin == C6_C && a.N == C6_C &&
         a.lda == C6_C && (a.ldc & 3) == 0 && (!a.R || (a.ldr & 3) == 0) && (!a.R2 || (a.ldr2 & 3) == 0) && (!a.C2 || (a.ldc2 & 3) == 0) &&
         a.taps >= 1 && a.dil >= 1 && (a.taps - 1) * a.dil <= C6_MAXHALO && a.pad >= 0 && a.pad <= (a.taps - 1) * a.dil &&
         a.nseg <= C6_MAXSEG && a.M >= disp().c64_min_rows && slab_rows_ok(a.M) &&
         (size_t)a.taps * C6_C * C6_C * 4 < 0x7ff00000ull && (a.in_act == ACT_NONE || (a.in_act == ACT_LRELU && a.in_slope > 0.f && a.in_slope < 1.f)) &&
         (a.act == ACT_NONE || a.act == ACT_LRELU);
}

static size_t c64_lds(int bm, const GemmArgs& a) {
  return (size_t)(((bm + (a.taps - 1) * a.dil) * C6_LDA + 3) & ~3) * sizeof(float) + (C6_MAXSEG + 2) * sizeof(int);
}
template <bool LRELU, int WM>
static int launch_c64_t(const GemmArgs& a, hipStream_t stream) {
  constexpr int BM = 64 * WM;
  const int slab_rows = BM + (a.taps - 1) * a.dil;
  const size_t lds = c64_lds(BM, a);
  SS_MAX_LDS_ONCE((&conv_c64_kernel<LRELU, WM>), 96 * 1024);
  SkWorkspace* st = nullptr;                       // (only for the device's CU count, cached per context)
  int rc = sk_workspace_acquire(stream, &st);
  if (rc != SS_OK) return rc;
  const int nseg = a.nseg > 0 ? a.nseg : 1;
  const long long max_blocks = (long long)cdiv(a.M, BM) + nseg;      // upper bound (per-segment round-up)
  const int occ = lds * 2 <= 158 * 1024 ? 2 : 1;   // resident workgroups per CU (LDS-limited; 164 registers per wave)
  const int grid = (int)std::min<long long>((long long)occ * st->cus, std::max<long long>(1, max_blocks));
  ProfRec rec{}; bool prof = false;
  rc = prof_begin(a, stream, 24, rec, prof);
  if (rc != SS_OK) return rc;
  hipLaunchKernelGGL((conv_c64_kernel<LRELU, WM>), dim3(grid), dim3(256), lds, stream, a, slab_rows);
  SS_LAUNCH_CHECK();
  return prof_end(stream, rec, prof);
}

int launch_conv_c64(const GemmArgs& a, hipStream_t stream) {
  if (!conv_c64_eligible(a)) return SS_ERR_ARG;
  // 256-row blocks when two workgroups' slabs fit a CU's LDS, else 192-row blocks (k = 11 at dilation 5: 83 KB -> 66 KB)
  static const int force_wm = getenv("SS_CONV_C64_WM") ? atoi(getenv("SS_CONV_C64_WM")) : 0;
  const bool wm4 = force_wm ? force_wm == 4 : 2 * c64_lds(256, a) <= 158 * 1024;
  if (a.in_act == ACT_LRELU) return wm4 ? launch_c64_t<true, 4>(a, stream) : launch_c64_t<true, 3>(a, stream);
  return wm4 ? launch_c64_t<false, 4>(a, stream) : launch_c64_t<false, 3>(a, stream);
}

}  // namespace ss
