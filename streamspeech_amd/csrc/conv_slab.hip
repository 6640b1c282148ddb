// Slab Conv1d for the narrow HiFi-GAN stages (C = 32 / 16 channels, 160x / 320x the frame rate;
// reference fairseq/models/text_to_speech/hifigan.py:52-172).
//
// These convs have K = taps*C <= 352 and N <= 32: an LDS-tiled GEMM spends its time on prologue
// and epilogue (3..11 k-steps per tile) and re-reads every input row once per tap from L2, while
// the arithmetic intensity (30-90 FLOP/B) sits right at the machine balance -- they should run at
// the HBM/MFMA corner.  Here:
//   * persistent workgroups (2 per CU); the whole weight matrix [N][taps*C] is loaded into LDS once
//     per workgroup (<= 45 KB, padded rows, conflict-free B fragments);
//   * per block of 128 output rows the input slab (128 + (taps-1)*dil rows x C) goes global -> LDS
//     ONCE (coalesced float4 loads, rows outside the utterance = the conv's zero padding, the input
//     leaky-ReLU applied here, once per element instead of once per tap) into rows padded to C+4
//     floats: tap-shifted ds_read_b128 A fragments are conflict-free and their addresses are linear
//     in the tap, so the MFMA loop carries ~2 VALU per 16 MFMAs (VALU issue costs matrix-core time);
//   * all taps are contracted out of LDS with v_mfma_f32_16x16x4_f32 (operands swapped, D = W.A^T,
//     so the epilogue is float4 along the channels);
//   * HBM traffic = input once (+ halo) + output once + residual operands: the algorithmic minimum.
// Blocks never straddle utterances (per-segment block table), so validity is uniform per block.
#include "gemm.hpp"

namespace ss {

using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int SL_BM = 128;
constexpr int SL_MAXSEG = 256;

template <int C, int N, bool LRELU>
__global__ __launch_bounds__(256, 2) void conv_slab_kernel(const GemmArgs p, const int slab_rows_max) {
#if __HIP_DEVICE_COMPILE__
  constexpr int BM = SL_BM;
  constexpr int Q = C / 4;                    // 16-B chunks per row
  constexpr int TM = BM / 64, TN = N / 16;    // wave tile: 32 rows x N
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int K = p.taps * C;
  const int LDW = K + 4;
  float* sW = smem;                                        // [N][K+4]
  float* sA = smem + ((N * LDW + 255) & ~255);             // slab [rows][C+4]
  int* s_blk = reinterpret_cast<int*>(sA + ((slab_rows_max * (C + 4) + 255) & ~255));   // block prefix per segment

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int r = lane & 15, g = lane >> 4;

  // weights -> LDS (once per workgroup)
  for (int idx = t; idx < N * (K / 4); idx += 256) {
    const int n = idx / (K / 4), k4 = idx - n * (K / 4);
    *reinterpret_cast<f32x4*>(sW + n * LDW + k4 * 4) = *reinterpret_cast<const f32x4*>(p.W + (size_t)n * K + k4 * 4);
  }
  // block table: s_blk[s] = first block of segment s, s_blk[nseg] = total
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

  constexpr int LDA = C + 4;                   // padded slab row (floats): 16 consecutive rows x 16 B hit 16 distinct slots
  const int slab_rows = BM + (p.taps - 1) * p.dil;
  const float slope = p.in_slope;

  // The slab of the NEXT block is fetched into registers before the current block is computed, so
  // the global-load latency hides under the MFMAs (a second LDS slab would halve the occupancy).
  constexpr int NP = Q;                        // float4 per thread: (BM + 128 halo rows) * Q / 256
  f32x4 pre[NP];
  int seg = 0, seg_lo = 0, seg_hi = 0, m0 = 0;
  auto locate = [&](int blk) {                 // blocks ascend per workgroup
    while (blk >= s_blk[seg + 1]) ++seg;
    seg_lo = p.nseg > 0 ? p.segs[4 * seg] : 0;
    seg_hi = seg_lo + (p.nseg > 0 ? p.segs[4 * seg + 1] : p.in_len);
    m0 = seg_lo + (blk - s_blk[seg]) * BM;     // first output row (packed coordinates)
  };
  auto prefetch = [&]() {
#pragma unroll
    for (int u = 0; u < NP; ++u) {
      const int idx = t + u * 256;             // Q consecutive threads read one 4C-byte row
      const int rho = idx / Q, c4 = idx - rho * Q;
      const int gin = m0 - p.pad + rho;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (rho < slab_rows && gin >= seg_lo && gin < seg_hi)
        v = *reinterpret_cast<const f32x4*>(p.A + (size_t)gin * p.lda + c4 * 4);
      pre[u] = v;
    }
  };
  int blk = blockIdx.x;
  if (blk < nblocks) { locate(blk); prefetch(); }
  for (; blk < nblocks; blk += gridDim.x) {
    const int cm0 = m0;
    const int m_hi = p.nseg > 0 ? seg_hi : min(seg_hi, p.M);
    __syncthreads();                                       // previous block's slab reads are done
#pragma unroll
    for (int u = 0; u < NP; ++u) {
      const int idx = t + u * 256;
      const int rho = idx / Q, c4 = idx - rho * Q;
      if (rho < slab_rows) {
        f32x4 v = pre[u];
        if (LRELU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * slope;
        }
        *reinterpret_cast<f32x4*>(sA + rho * LDA + c4 * 4) = v;
      }
    }
    __syncthreads();
    if (blk + (int)gridDim.x < nblocks) { locate(blk + gridDim.x); prefetch(); }

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const float* pa = sA + (wave * 32 + r) * LDA + 4 * g;  // + i*16*LDA + tap*dil*LDA + cc*16
    const float* pw = sW + r * LDW + 4 * g;                // + j*16*LDW + tap*C + cc*16
    const int a_step = p.dil * LDA;
    for (int tap = 0; tap < p.taps; ++tap) {
#pragma unroll
      for (int cc = 0; cc < C / 16; ++cc) {
        f32x4 af[TM], bf[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const f32x4*>(pa + i * 16 * LDA + cc * 16);
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const f32x4*>(pw + j * 16 * LDW + cc * 16);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[j][e], af[i][e], acc[i][j], 0, 0, 0);   // D = W.A^T
      }
      pa += a_step;
      pw += C;
    }

    // epilogue: lane holds 4 consecutive channels (4g..4g+3 of n-tile j) of row r (see conv_sk.hip)
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int m = cm0 + wave * 32 + i * 16 + r;
      if (m >= m_hi) continue;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int n = j * 16 + g * 4;
        f32x4 v = acc[i][j];
        if (p.bias) {
          const f32x4 b = *reinterpret_cast<const f32x4*>(p.bias + n);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += b[e];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          switch (p.act) {
            case ACT_SILU: v[e] = v[e] / (1.0f + expf(-v[e])); break;
            case ACT_RELU: v[e] = fmaxf(v[e], 0.f); break;
            case ACT_TANH: v[e] = tanhf(v[e]); break;
            case ACT_LRELU: v[e] = v[e] > 0.f ? v[e] : v[e] * p.act_slope; break;
            default: break;
          }
          v[e] *= p.alpha;
        }
        if (p.R) {
          const f32x4 rr = *reinterpret_cast<const f32x4*>(p.R + (size_t)m * p.ldr + n);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += rr[e];
        }
        if (p.R2) {
          const f32x4 rr = *reinterpret_cast<const f32x4*>(p.R2 + (size_t)m * p.ldr2 + n);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = rr[e] + v[e];
        }
        if (p.div > 0.f) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = v[e] / p.div;
        }
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

bool conv_slab_eligible(const GemmArgs& a) {
  return a.same_rows && a.stride == 1 && a.chunk == 0 && !a.glu && !a.ln_g && (a.Cin == 32 || a.Cin == 16) &&
         (a.N == 32 || a.N == 16) && a.lda == a.Cin && (a.ldc & 3) == 0 && (!a.R || (a.ldr & 3) == 0) &&
         (!a.R2 || (a.ldr2 & 3) == 0) && (!a.C2 || (a.ldc2 & 3) == 0) && a.taps >= 1 && a.taps * a.Cin <= 512 &&
         (a.taps - 1) * a.dil <= 128 &&   // slab <= 256 rows (register prefetch budget)
         a.nseg <= SL_MAXSEG && a.M > 0 &&
         slab_rows_ok(a.M) &&
         (a.in_act == ACT_NONE || a.in_act == ACT_LRELU);
}

static int slab_cus(int& cus) {           // CU count of the CURRENT device, read once per device (thread-safe)
  static std::mutex mu;
  static int n[128] = {0};
  int dev = 0;
  SS_HIP_CHECK(hipGetDevice(&dev));
  if (dev < 0 || dev >= 128) return SS_ERR_ARG;
  std::lock_guard<std::mutex> lk(mu);
  if (n[dev] == 0) {
    int v = 0;
    SS_HIP_CHECK(hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev));
    n[dev] = v > 0 ? v : 256;
  }
  cus = n[dev];
  return SS_OK;
}

template <int C, int N, bool LRELU>
static int launch_slab_t(const GemmArgs& a, hipStream_t stream, int cls) {
  const int K = a.taps * C;
  const int slab_rows = SL_BM + (a.taps - 1) * a.dil;
  const size_t lds = ((size_t)((N * (K + 4) + 255) & ~255) + (size_t)((slab_rows * (C + 4) + 255) & ~255)) * sizeof(float) +
                     (SL_MAXSEG + 2) * sizeof(int);
  if (lds > 80 * 1024) return SS_ERR_ARG;
  SS_MAX_LDS_ONCE((&conv_slab_kernel<C, N, LRELU>), 80 * 1024);
  int g_slab_cus = 0;
  { int rc_cus = slab_cus(g_slab_cus); if (rc_cus != SS_OK) return rc_cus; }
  const int nseg = a.nseg > 0 ? a.nseg : 1;
  const long long max_blocks = (long long)cdiv(a.M, SL_BM) + nseg;      // upper bound (per-segment round-up)
  const int occ = (int)std::max<size_t>(2, std::min<size_t>(6, (150 * 1024) / lds));   // resident workgroups per CU (LDS-limited)
  const int grid = (int)std::min<long long>((long long)occ * g_slab_cus, std::max<long long>(1, max_blocks));
  ProfRec rec{}; bool prof = false;
  int rc = prof_begin(a, stream, cls, rec, prof);
  if (rc != SS_OK) return rc;
  hipLaunchKernelGGL((conv_slab_kernel<C, N, LRELU>), dim3(grid), dim3(256), lds, stream, a, slab_rows);
  SS_LAUNCH_CHECK();
  return prof_end(stream, rec, prof);
}

int launch_conv_slab(const GemmArgs& a, hipStream_t stream) {
  if (!conv_slab_eligible(a)) return SS_ERR_ARG;
  const bool lr = a.in_act == ACT_LRELU;
#define SS_SLAB(C_, N_, CLS_) \
  if (a.Cin == C_ && a.N == N_) return lr ? launch_slab_t<C_, N_, true>(a, stream, CLS_) : launch_slab_t<C_, N_, false>(a, stream, CLS_);
  SS_SLAB(32, 32, 16) SS_SLAB(16, 16, 17) SS_SLAB(32, 16, 16) SS_SLAB(16, 32, 17)
#undef SS_SLAB
  return SS_ERR_ARG;
}

// =================================================================================================
// Fused resblock half: y = conv2(lrelu(conv1(lrelu(x)) + b1)) + b2 + x  (hifigan.py:95-102, one of the
// three (dilated conv, plain conv, residual add) pairs of a ResBlock) for the narrow stages, where both
// weight matrices fit in LDS next to the slabs.  Per block of 128 output rows: the activated input slab
// (128 + halo of both convs) is staged once, conv1 is contracted into an LDS `mid` slab (bias, leaky-ReLU,
// zero outside the utterance = conv2's padding), conv2 reads it back -- the intermediate tensor never
// goes to HBM (5 tensor passes per pair -> 2: read x, write y).  Same per-element arithmetic as the two
// separate launches (same tap/channel order inside each conv), so results are bit-identical to them.
// =================================================================================================
// NT (1 or 2) 16-row tiles of one conv contracted together out of LDS slabs: two tiles give two independent
// accumulator chains per n-tile (a single chain is bound by the MFMA dependent-issue latency) and share the
// weight fragments.
template <int C, int NT>
__device__ __forceinline__ void slab_tiles(const float* __restrict__ slab, int lda, const float* __restrict__ sW, int ldw,
                                           int taps, int tap_rows, int r, int g, const int (&tile)[2],
                                           f32x4 (&acc)[2][C / 16]) {
  constexpr int TN = C / 16;
#pragma unroll
  for (int u = 0; u < NT; ++u)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[u][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float* pa0 = slab + (tile[0] * 16 + r) * lda + 4 * g;
  const float* pa1 = slab + (tile[NT - 1] * 16 + r) * lda + 4 * g;
  const float* pw = sW + r * ldw + 4 * g;
  for (int tap = 0; tap < taps; ++tap) {
#pragma unroll
    for (int cc = 0; cc < C / 16; ++cc) {
      f32x4 af[2];
      af[0] = *reinterpret_cast<const f32x4*>(pa0 + cc * 16);
      if (NT == 2) af[1] = *reinterpret_cast<const f32x4*>(pa1 + cc * 16);
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const f32x4 bf = *reinterpret_cast<const f32x4*>(pw + j * 16 * ldw + cc * 16);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int u = 0; u < NT; ++u)
            acc[u][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[e], af[u][e], acc[u][j], 0, 0, 0);
      }
    }
    pa0 += tap_rows * lda;
    pa1 += tap_rows * lda;
    pw += C;
  }
}

struct PairArgs {
  const float* A = nullptr; int lda = 0;            // x: the un-activated residual stream
  const float* W1 = nullptr; const float* b1 = nullptr;
  const float* W2 = nullptr; const float* b2 = nullptr;
  float* C = nullptr; int ldc = 0;                  // y (must not alias A: neighbouring blocks read A's halo)
  const float* R2 = nullptr; int ldr2 = 0; float div = 0.f;
  float* C2 = nullptr; int ldc2 = 0; float c2_slope = 0.1f;
  int taps = 3, dil = 1, M = 0, in_len = 0;
  float slope = 0.1f;
  const int* segs = nullptr; int nseg = 0;
};

template <int C>
__global__ __launch_bounds__(256, 2) void conv_pair_kernel(const PairArgs p, const int slab_rows) {
#if __HIP_DEVICE_COMPILE__
  constexpr int BM = SL_BM, MT = BM / 16 + 1;  // 9 mid tiles (144 rows >= 128 + 2*h2), 8 output tiles
  constexpr int Q = C / 4, TN = C / 16, LDA = C + 4;
  constexpr int NP = (194 * Q + 255) / 256;    // slab float4 per thread (<= 144 + 50 rows)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int K = p.taps * C, LDW = K + 4;
  float* sW1 = smem;
  float* sW2 = sW1 + ((C * LDW + 255) & ~255);
  float* sA = sW2 + ((C * LDW + 255) & ~255);                    // [slab_rows][LDA]
  float* sM = sA + ((slab_rows * LDA + 255) & ~255);             // [144][LDA]
  int* s_blk = reinterpret_cast<int*>(sM + ((MT * 16 * LDA + 255) & ~255));

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int r = lane & 15, g = lane >> 4;
  for (int idx = t; idx < C * (K / 4); idx += 256) {
    const int n = idx / (K / 4), k4 = idx - n * (K / 4);
    *reinterpret_cast<f32x4*>(sW1 + n * LDW + k4 * 4) = *reinterpret_cast<const f32x4*>(p.W1 + (size_t)n * K + k4 * 4);
    *reinterpret_cast<f32x4*>(sW2 + n * LDW + k4 * 4) = *reinterpret_cast<const f32x4*>(p.W2 + (size_t)n * K + k4 * 4);
  }
  const int nseg = p.nseg > 0 ? p.nseg : 1;
  if (t == 0) {
    int acc = 0;
    for (int s = 0; s < nseg; ++s) {
      s_blk[s] = acc;
      acc += ((p.nseg > 0 ? p.segs[4 * s + 1] : p.M) + BM - 1) / BM;
    }
    s_blk[nseg] = acc;
  }
  __syncthreads();
  const int nblocks = s_blk[nseg];
  const int h1 = p.dil * (p.taps - 1) / 2, h2 = (p.taps - 1) / 2;
  const float slope = p.slope;

  f32x4 pre[NP];
  int seg = 0, seg_lo = 0, seg_hi = 0, m0 = 0;
  auto locate = [&](int blk) {
    while (blk >= s_blk[seg + 1]) ++seg;
    seg_lo = p.nseg > 0 ? p.segs[4 * seg] : 0;
    seg_hi = seg_lo + (p.nseg > 0 ? p.segs[4 * seg + 1] : p.in_len);
    m0 = seg_lo + (blk - s_blk[seg]) * BM;
  };
  auto prefetch = [&]() {
#pragma unroll
    for (int u = 0; u < NP; ++u) {
      const int idx = t + u * 256;
      const int rho = idx / Q, c4 = idx - rho * Q;
      const int gin = m0 - h2 - h1 + rho;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (rho < slab_rows && gin >= seg_lo && gin < seg_hi)
        v = *reinterpret_cast<const f32x4*>(p.A + (size_t)gin * p.lda + c4 * 4);
      pre[u] = v;
    }
  };
  int blk = blockIdx.x;
  if (blk < nblocks) { locate(blk); prefetch(); }
  for (; blk < nblocks; blk += gridDim.x) {
    const int cm0 = m0, clo = seg_lo, chi = seg_hi;
    __syncthreads();                                       // previous block's conv2 is done with both slabs
#pragma unroll
    for (int u = 0; u < NP; ++u) {
      const int idx = t + u * 256;
      const int rho = idx / Q, c4 = idx - rho * Q;
      if (rho < slab_rows) {
        f32x4 v = pre[u];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * slope;
        *reinterpret_cast<f32x4*>(sA + rho * LDA + c4 * 4) = v;
      }
    }
    __syncthreads();
    if (blk + (int)gridDim.x < nblocks) { locate(blk + gridDim.x); prefetch(); }

    // ---- conv1 (dilated) -> mid slab ----
    auto conv1_out = [&](int mt, const f32x4 (&a)[TN]) {   // bias, leaky-ReLU, zero outside the utterance -> mid slab
      const int gm = cm0 - h2 + mt * 16 + r;               // global row of this lane's mid row
      const bool in_utt = gm >= clo && gm < chi;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(p.b1 + j * 16 + 4 * g);
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float x = a[j][e] + b[e];
          v[e] = in_utt ? (x > 0.f ? x : x * slope) : 0.f;
        }
        *reinterpret_cast<f32x4*>(sM + (mt * 16 + r) * LDA + j * 16 + 4 * g) = v;
      }
    };
    {
      f32x4 acc[2][TN];
      const int t2[2] = {wave, wave + 4};                  // mid tiles w and w+4 together, the 9th tile on the last wave
      slab_tiles<C, 2>(sA, LDA, sW1, LDW, p.taps, p.dil, r, g, t2, acc);
      conv1_out(t2[0], acc[0]);
      conv1_out(t2[1], acc[1]);
      if (wave == 3) {
        const int t1[2] = {MT - 1, MT - 1};
        slab_tiles<C, 1>(sA, LDA, sW1, LDW, p.taps, p.dil, r, g, t1, acc);
        conv1_out(t1[0], acc[0]);
      }
    }
    __syncthreads();

    // ---- conv2 (dilation 1) over the mid slab + residual ----
    f32x4 acc2[2][TN];
    const int o2[2] = {wave, wave + 4};
    slab_tiles<C, 2>(sM, LDA, sW2, LDW, p.taps, 1, r, g, o2, acc2);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int m = cm0 + o2[u] * 16 + r;
      if (m >= chi) continue;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int n = j * 16 + 4 * g;
        const f32x4 b = *reinterpret_cast<const f32x4*>(p.b2 + n);
        const f32x4 xr = *reinterpret_cast<const f32x4*>(p.A + (size_t)m * p.lda + n);
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (acc2[u][j][e] + b[e]) + xr[e];
        if (p.R2) {
          const f32x4 rr = *reinterpret_cast<const f32x4*>(p.R2 + (size_t)m * p.ldr2 + n);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = rr[e] + v[e];
        }
        if (p.div > 0.f) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = v[e] / p.div;
        }
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

size_t conv_pair_lds(int C, int taps, int dil) {
  const int K = taps * C;
  const int slab_rows = 144 + (taps - 1) * dil;
  auto r256 = [](size_t x) { return (x + 255) & ~(size_t)255; };
  return (2 * r256((size_t)C * (K + 4)) + r256((size_t)slab_rows * (C + 4)) + r256((size_t)144 * (C + 4))) * sizeof(float) +
         (SL_MAXSEG + 2) * sizeof(int);
}

bool conv_pair_eligible(int C, int taps, int dil, int lda, int ldc, int nseg, long long M) {
  return (C == 16 || C == 32) && lda == C && (ldc & 3) == 0 && (taps & 1) == 1 && taps * C <= 512 && (taps - 1) * dil <= 50 &&
         nseg <= SL_MAXSEG && M >= 2048 && conv_pair_lds(C, taps, dil) <= 72 * 1024 &&
         slab_rows_ok(M);
}

template <int C>
static int launch_pair_t(const PairArgs& a, hipStream_t stream) {
  const size_t lds = conv_pair_lds(C, a.taps, a.dil);
  SS_MAX_LDS_ONCE((&conv_pair_kernel<C>), 72 * 1024);
  int g_slab_cus = 0;
  { int rc_cus = slab_cus(g_slab_cus); if (rc_cus != SS_OK) return rc_cus; }
  const int nseg = a.nseg > 0 ? a.nseg : 1;
  const long long max_blocks = (long long)cdiv(a.M, SL_BM) + nseg;
  const int occ = (int)std::max<size_t>(2, std::min<size_t>(4, (150 * 1024) / lds));
  const int grid = (int)std::min<long long>((long long)occ * g_slab_cus, std::max<long long>(1, max_blocks));
  // profiler class of the slab kernels (C = 32 -> 16, C = 16 -> 17); FLOPs of both convs
  GemmArgs ga;
  ga.M = a.M; ga.N = C; ga.Cin = C; ga.taps = a.taps; ga.in_len = a.in_len; ga.algo_flops = 4.0 * (double)a.M * C * C * a.taps;
  ProfRec rec{}; bool prof = false;
  int rc = prof_begin(ga, stream, C == 32 ? 16 : 17, rec, prof);
  if (rc != SS_OK) return rc;
  hipLaunchKernelGGL((conv_pair_kernel<C>), dim3(grid), dim3(256), lds, stream, a, 144 + (a.taps - 1) * a.dil);
  SS_LAUNCH_CHECK();
  return prof_end(stream, rec, prof);
}

int launch_conv_pair(const float* A, int lda, const float* W1, const float* b1, const float* W2, const float* b2, float* Cc,
                     int ldc, const float* R2, int ldr2, float div, float* C2, int ldc2, float c2_slope, int C, int taps,
                     int dil, int M, int in_len, float slope, const int* segs, int nseg, hipStream_t stream) {
  if (!conv_pair_eligible(C, taps, dil, lda, ldc, nseg, M) || A == Cc) return SS_ERR_ARG;
  PairArgs a;
  a.A = A; a.lda = lda; a.W1 = W1; a.b1 = b1; a.W2 = W2; a.b2 = b2; a.C = Cc; a.ldc = ldc; a.R2 = R2; a.ldr2 = ldr2;
  a.div = div; a.C2 = C2; a.ldc2 = ldc2; a.c2_slope = c2_slope; a.taps = taps; a.dil = dil; a.M = M; a.in_len = in_len;
  a.slope = slope; a.segs = segs; a.nseg = nseg;
  return C == 32 ? launch_pair_t<32>(a, stream) : launch_pair_t<16>(a, stream);
}
}  // namespace ss
