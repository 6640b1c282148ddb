// Fused HiFi-GAN ResBlock for the narrow generator stages (C = 32 / 16 channels at 160x / 320x the frame rate):
//
//   for d in (1, 3, 5):   xt = conv2_d(lrelu(conv1_d(lrelu(x)))) ;  x = xt + x          (hifigan.py:95-102)
//   out = [R2 +] x [/ div]                                                              (MRF sum / mean, hifigan.py:159-165)
//
// ONE persistent launch per ResBlock: the three (dilated conv, plain conv, residual add) pairs run back to back on a
// block of rows without any intermediate leaving the CU.  The multi-launch form moved every intermediate of the narrow
// stages through HBM 5-6 times per pair (conv_slab) or twice (conv_pair, k = 3 only) and sat at MfmaUtil 40-44 % and
// 0.2-0.3 of HBM, bound by neither; here a ResBlock reads x once (+ halo) and writes its result once.
//
// Per block of BM = 16*NTILE - 2H output rows (H = sum over the pairs of both convs' half-widths = 6 (k-1): the
// ResBlock's receptive half-width; the halo rows are recomputed, that is the price of the fusion):
//   * slab coordinates: row rho <-> global row m0 - H + rho, 16-row tiles; tile tau belongs to wave tau % 8 for the whole
//     block, so the RAW residual stream x of a tile lives in that wave's REGISTERS across the three pairs (the conv2
//     epilogue of a pair reads and rewrites it); LDS holds only what the MFMAs read: the activated slab lrelu(x)
//     (zero outside the utterance = every conv's padding) and the conv1 -> conv2 intermediate `mid`;
//   * the weights of the conv being contracted sit in LDS ([C][k C + 4], conflict-free fragments); the next conv's
//     matrix is fetched from L2 into registers under the MFMA loop and swapped in between two barriers (six matrices
//     do not fit next to the slabs: 6 x 45 KB at C = 32, k = 11);
//   * the valid range shrinks by (h1 + h2) per pair; only tiles that intersect it are contracted (the first / last
//     slot of a wave drops out: four statically-unrolled variants of the loop);
//   * the next block's x rows are requested into registers during the last conv of the current block.
// Arithmetic per element (tap -> 16-channel group -> k order inside v_mfma_f32_16x16x4_f32, bias, leaky-ReLU, residual,
// MRF add, mean) is the one conv_slab_kernel / conv_pair_kernel perform: results are bit-identical to the multi-launch
// form (tests/test_batch_gpu.py).
#include "gemm.hpp"

#include <type_traits>

namespace ss {

using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int RB_NW = 8;          // waves per workgroup (2 per SIMD)
constexpr int RB_MAXSEG = 256;

struct ResblockArgs {
  const float* X = nullptr; int ldx = 0;            // stage input x (raw residual stream) [M][C]
  const float* W1[3] = {nullptr, nullptr, nullptr}; const float* B1[3] = {nullptr, nullptr, nullptr};
  const float* W2[3] = {nullptr, nullptr, nullptr}; const float* B2[3] = {nullptr, nullptr, nullptr};
  int dil[3] = {1, 3, 5};
  float* Y = nullptr; int ldy = 0;                  // out (may alias R2, must not alias X)
  const float* R2 = nullptr; int ldr2 = 0; float div = 0.f;
  int M = 0;
  const int* segs = nullptr; int nseg = 0;          // {start, len, -, -} per utterance (stride 4)
  float slope = 0.1f;
};

// Work unit of a wave = (16-row tile, 16-channel output block).  C = 32: wave w contracts output block w & 1 of the tiles
// tau = (w >> 1) (mod 4) -- half-tile units keep the four SIMDs (waves w and w + 4 share one) within one unit of each other
// when the valid range drops tiles at either end; C = 16: one output block, tiles tau = w (mod 8).
template <int C, int TAPS> struct RbGeom {
  static constexpr int LDA = C + 4, K = TAPS * C, LDW = K + 4, HT = (TAPS - 1) / 2, CC = C / 16;
  static constexpr int TSTRIDE = C == 32 ? 4 : 8;
  // slab tiles: bounded by 160 KB of LDS (two slabs with one margin tile on either side + the weight buffer(s))
  static constexpr int NTILE = C == 32 ? (TAPS >= 11 ? 23 : 24) : 40;
  static constexpr int SLOTS = (NTILE + TSTRIDE - 1) / TSTRIDE;
  static constexpr int RX = NTILE * 16;
  static constexpr int W_FLOATS = (C * LDW + 3) & ~3;                       // 16-B aligned pieces (every byte counts at C = 32, k = 11)
  static constexpr int SLAB_FLOATS = ((NTILE + 2) * 16 * LDA + 3) & ~3;
  static constexpr int BIAS_FLOATS = 6 * C;
  // two weight buffers where they fit: the next conv's matrix is then written while the current one is being read and
  // a conv costs ONE workgroup barrier instead of two
  static constexpr bool WDB = (size_t)(2 * W_FLOATS + 2 * SLAB_FLOATS + BIAS_FLOATS) * 4 + (RB_MAXSEG + 2) * 4 <= 160 * 1024;
  static constexpr int NWB = WDB ? 2 : 1;
  static constexpr size_t LDS_BYTES = (size_t)(NWB * W_FLOATS + 2 * SLAB_FLOATS + BIAS_FLOATS) * sizeof(float) + (RB_MAXSEG + 2) * sizeof(int);
  static constexpr int WV4 = (C * (K / 4) + RB_NW * 64 - 1) / (RB_NW * 64);   // weight float4 per thread
};

// One conv of the ResBlock for the wave's unit slots [S0, S1): acc = sum_tap sum_cc sum_e  W . src^T out of LDS.
// Software-pipelined at half-step granularity: the units are split into two groups; while one group's MFMAs of step n
// issue, the other group's fragments (of step n, then of step n + 1) are in flight -- two 3-fragment register sets instead
// of two full ones (the 512-thread workgroup leaves 256 registers per wave).
// `drows` = rows per tap (the dilation); the tap-0 row of output row rho is rho - HT * drows.
template <int C, int TAPS, int S0, int S1, class Epi>
__device__ __forceinline__ void rb_conv(const float* __restrict__ src, const float* __restrict__ sW, int drows, int tw, int jw,
                                        int r, int g, Epi&& epi) {
  using G = RbGeom<C, TAPS>;
  constexpr int LDA = G::LDA, LDW = G::LDW, NS = S1 - S0, CC = G::CC, TS = G::TSTRIDE;
  constexpr int NSTEP = TAPS * CC, NPAIR = NSTEP / 2;
  constexpr bool ODD = (NSTEP & 1) != 0;
  constexpr int NA = (NS + 1) / 2, NB = NS - NA;           // unit groups: [0, NA) and [NA, NS)
  if constexpr (NS > 0) {
    f32x4 acc[NS];
#pragma unroll
    for (int u = 0; u < NS; ++u) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* pa = src + ((S0 * TS + tw) * 16 + r - G::HT * drows) * LDA + 4 * g;
    const float* pw = sW + (jw * 16 + r) * LDW + 4 * g;
    const int a_step = drows * LDA;
    // step n -> (tap, cc) = (n / CC, n % CC): A advances 16 floats inside a tap and a_step between taps, W 16 floats per step
    const int off1 = CC == 2 ? 16 : a_step;                  // step n + 1 relative to step n (n even)
    const int off2 = CC == 2 ? a_step : 2 * a_step;          // step n + 2
    f32x4 ga[NA], gb[NB > 0 ? NB : 1], b0, b1;
    auto loadA = [&](const float* qa) {
#pragma unroll
      for (int u = 0; u < NA; ++u) ga[u] = *reinterpret_cast<const f32x4*>(qa + u * TS * 16 * LDA);
    };
    auto loadB = [&](const float* qa) {
#pragma unroll
      for (int u = 0; u < NB; ++u) gb[u] = *reinterpret_cast<const f32x4*>(qa + (NA + u) * TS * 16 * LDA);
    };
    auto mmaA = [&](const f32x4& bf) {
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int u = 0; u < NA; ++u)
          acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[e], ga[u][e], acc[u], 0, 0, 0);   // D = W . A^T
    };
    auto mmaB = [&](const f32x4& bf) {
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int u = 0; u < NB; ++u)
          acc[NA + u] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[e], gb[u][e], acc[NA + u], 0, 0, 0);
    };
    loadA(pa);
    b0 = *reinterpret_cast<const f32x4*>(pw);
    // two steps per trip; the trip that has nothing left to request is peeled so that the loop body stays ONE basic block
    auto trip = [&](auto next) {
      loadB(pa);                                             // step n, second group
      mmaA(b0);
      loadA(pa + off1);                                      // step n + 1, first group
      b1 = *reinterpret_cast<const f32x4*>(pw + 16);
      mmaB(b0);
      loadB(pa + off1);
      mmaA(b1);
      if constexpr (decltype(next)::value) {                 // step n + 2
        loadA(pa + off2);
        b0 = *reinterpret_cast<const f32x4*>(pw + 32);
      }
      mmaB(b1);
      pa += off2;
      pw += 32;
    };
    constexpr int NFULL = ODD ? NPAIR : NPAIR - 1;           // trips that are followed by another step
#pragma unroll 1
    for (int it = 0; it < NFULL; ++it) trip(std::true_type{});
    if constexpr (ODD) {
      loadB(pa);
      mmaA(b0);
      mmaB(b0);
    } else {
      trip(std::false_type{});
    }
#pragma unroll
    for (int u = 0; u < NS; ++u) epi(S0 + u, acc[u]);
  }
}

template <int C, int TAPS>
__global__ __launch_bounds__(RB_NW * 64) void resblock_fused_kernel(const ResblockArgs p) {
#if __HIP_DEVICE_COMPILE__
  using G = RbGeom<C, TAPS>;
  constexpr int LDA = G::LDA, K = G::K, LDW = G::LDW, HT = G::HT, NTILE = G::NTILE, SLOTS = G::SLOTS, RX = G::RX, TS = G::TSTRIDE;
  constexpr int NT = RB_NW * 64, WV4 = G::WV4;
  constexpr bool WDB = G::WDB;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sWb = smem;                                                   // NWB x [C][LDW]
  float* sX = sWb + G::NWB * G::W_FLOATS + 16 * LDA;                   // activated x slab, row 0 after one margin tile
  float* sM = sWb + G::NWB * G::W_FLOATS + G::SLAB_FLOATS + 16 * LDA;  // mid slab
  float* sB = sWb + G::NWB * G::W_FLOATS + 2 * G::SLAB_FLOATS;         // biases [6][C]: conv c = 2 * pair + (conv2 ? 1 : 0)
  int* s_blk = reinterpret_cast<int*>(sB + G::BIAS_FLOATS);

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int tw = C == 32 ? (wave >> 1) : wave;               // tile class of this wave: tiles tw, tw + TS, ...
  const int jw = C == 32 ? (wave & 1) : 0;                   // 16-channel output block of this wave
  const int r = lane & 15, g = lane >> 4;
  const int n_l = jw * 16 + 4 * g;                           // this lane's 4 output channels
  const float slope = p.slope;
  const int H = HT * (p.dil[0] + p.dil[1] + p.dil[2] + 3);
  const int BM = RX - 2 * H;

  // block table: s_blk[s] = first block of utterance s
  const int nseg = p.nseg > 0 ? p.nseg : 1;
  if (t == 0) {
    int acc = 0;
    for (int s = 0; s < nseg; ++s) {
      s_blk[s] = acc;
      acc += ((p.nseg > 0 ? p.segs[4 * s + 1] : p.M) + BM - 1) / BM;
    }
    s_blk[nseg] = acc;
  }
  // margins and slabs start as zeros (rows no valid output ever depends on, but keep them finite)
  for (int i = t; i < 2 * G::SLAB_FLOATS; i += NT) sWb[G::NWB * G::W_FLOATS + i] = 0.f;
  if (t < 6 * C) {
    const int c = t / C, n = t - c * C;
    sB[t] = ((c & 1) ? p.B2[c >> 1] : p.B1[c >> 1])[n];
  }
  __syncthreads();
  const int nblocks = s_blk[nseg];

  // weights of conv `c` (0..5: pair c/2, conv1 / conv2) global -> registers -> LDS buffer c % NWB
  f32x4 wreg[WV4];
  auto w_fetch = [&](int c) {
    const float* Wg = (c & 1) ? p.W2[c >> 1] : p.W1[c >> 1];
#pragma unroll
    for (int u = 0; u < WV4; ++u) {
      const int idx = t + u * NT;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (idx < C * (K / 4)) v = *reinterpret_cast<const f32x4*>(Wg + (size_t)idx * 4);
      wreg[u] = v;
    }
  };
  auto w_commit = [&](int c) {
    float* sW = sWb + (WDB ? (c & 1) : 0) * G::W_FLOATS;
#pragma unroll
    for (int u = 0; u < WV4; ++u) {
      const int idx = t + u * NT;
      if (idx < C * (K / 4)) {
        const int n = idx / (K / 4), k4 = idx - n * (K / 4);
        *reinterpret_cast<f32x4*>(sW + n * LDW + k4 * 4) = wreg[u];
      }
    }
  };
  // hand-over between two convs: the next conv's matrix goes into LDS and everything the finished conv wrote
  // (slab rows) becomes visible.  Two buffers: the other buffer was last read one conv ago, behind a barrier -> one barrier.
  auto conv_done = [&](int c_next) {
    if constexpr (WDB) {
      w_commit(c_next);
      __syncthreads();
    } else {
      __syncthreads();                                      // everyone is done reading the single buffer
      w_commit(c_next);
      __syncthreads();
    }
  };

  // block geometry + the raw x values of this wave's units (epilogue layout: lane (r, g) holds row tau*16 + r,
  // channels n_l .. n_l + 3)
  int seg = 0, seg_lo = 0, seg_hi = 0, m0 = 0;
  auto locate = [&](int blk) {                 // blocks ascend per workgroup
    while (blk >= s_blk[seg + 1]) ++seg;
    seg_lo = p.nseg > 0 ? p.segs[4 * seg] : 0;
    seg_hi = seg_lo + (p.nseg > 0 ? p.segs[4 * seg + 1] : p.M);
    m0 = seg_lo + (blk - s_blk[seg]) * BM;
  };
  f32x4 pre[SLOTS], xr[SLOTS];
  auto x_fetch = [&]() {
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
      const int tau = s * TS + tw;
      const int gm = m0 - H + tau * 16 + r;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (tau < NTILE && gm >= seg_lo && gm < seg_hi) v = *reinterpret_cast<const f32x4*>(p.X + (size_t)gm * p.ldx + n_l);
      pre[s] = v;
    }
  };

  int blk = blockIdx.x;
  if (blk < nblocks) {
    locate(blk);
    x_fetch();
    w_fetch(0);
    w_commit(0);
  }
  for (; blk < nblocks; blk += gridDim.x) {
    const int cm0 = m0, clo = seg_lo, chi = seg_hi;
    // ---- stage: raw x -> registers, lrelu(x) -> slab (rows outside the utterance were fetched as zeros) ----
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
      const int tau = s * TS + tw;
      const f32x4 v = pre[s];
      xr[s] = v;
      if (tau < NTILE) {
        f32x4 a;
#pragma unroll
        for (int e = 0; e < 4; ++e) a[e] = v[e] > 0.f ? v[e] : v[e] * slope;
        *reinterpret_cast<f32x4*>(sX + (tau * 16 + r) * LDA + n_l) = a;
      }
    }
    const bool more = blk + (int)gridDim.x < nblocks;
    __syncthreads();                                       // slab + weights of conv 0 visible

    int vlo = 0, vhi = RX;                                 // valid rows of the x slab entering the pair
#pragma unroll 1
    for (int pair = 0; pair < 3; ++pair) {
      const int d = p.dil[pair];
      const int h1 = HT * d, h2 = HT;
      const int tl = (SLOTS - 1) * TS + tw;                // this wave's last slot's tile
      // ================= conv1 (dilated): activated x slab -> mid slab =================
      {
        const int lo = vlo + h1, hi = vhi - h1;            // valid mid rows
        const int s0 = (tw * 16 + 16 > lo) ? 0 : 1;        // slot 0 of this wave intersects the range?
        const int s1 = (tl < NTILE && tl * 16 < hi) ? SLOTS : SLOTS - 1;
        w_fetch(2 * pair + 1);                             // conv2's matrix, under the MFMAs
        const float* sW = sWb + (WDB ? 0 : 0) * G::W_FLOATS;   // conv c = 2 * pair: buffer 0
        const f32x4 b = *reinterpret_cast<const f32x4*>(sB + (2 * pair) * C + n_l);
        auto epi = [&](int s, const f32x4& a) {
          // per-lane addresses derive from a lane id the compiler cannot see through: otherwise it computes every slot's
          // epilogue address BEFORE the MFMA loop and carries them through it (the 256-register budget spills)
          int le = lane;
          asm volatile("" : "+v"(le));
          const int rho = (s * TS + tw) * 16 + (le & 15);
          const int gm = cm0 - H + rho;
          const bool in_utt = gm >= clo && gm < chi;
          f32x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float x = a[e] + b[e];
            v[e] = in_utt ? (x > 0.f ? x : x * slope) : 0.f;
          }
          *reinterpret_cast<f32x4*>(sM + rho * LDA + jw * 16 + 4 * (le >> 4)) = v;
        };
        if (s0 == 0 && s1 == SLOTS) rb_conv<C, TAPS, 0, SLOTS>(sX, sW, d, tw, jw, r, g, epi);
        else if (s0 == 0) rb_conv<C, TAPS, 0, SLOTS - 1>(sX, sW, d, tw, jw, r, g, epi);
        else if (s1 == SLOTS) rb_conv<C, TAPS, 1, SLOTS>(sX, sW, d, tw, jw, r, g, epi);
        else rb_conv<C, TAPS, 1, SLOTS - 1>(sX, sW, d, tw, jw, r, g, epi);
        vlo = lo; vhi = hi;
      }
      conv_done(2 * pair + 1);                             // mid complete, conv2's matrix in place
      // ================= conv2 (dilation 1): mid slab -> x (+ residual) =================
      {
        const int lo = vlo + h2, hi = vhi - h2;            // valid rows of the new x
        const int s0 = (tw * 16 + 16 > lo) ? 0 : 1;
        const int s1 = (tl < NTILE && tl * 16 < hi) ? SLOTS : SLOTS - 1;
        const bool last = pair == 2;
        if (!last) w_fetch(2 * pair + 2);                  // next pair's conv1
        else {
          w_fetch(0);                                      // next block starts with conv 0 again
          if (more) { locate(blk + gridDim.x); x_fetch(); }
        }
        const float* sW = sWb + (WDB ? 1 : 0) * G::W_FLOATS;   // conv c = 2 * pair + 1: buffer 1
        const f32x4 b = *reinterpret_cast<const f32x4*>(sB + (2 * pair + 1) * C + n_l);
        auto epi = [&](int s, const f32x4& a) {
          int le = lane;
          asm volatile("" : "+v"(le));
          const int rho = (s * TS + tw) * 16 + (le & 15);
          const int n_e = jw * 16 + 4 * (le >> 4);
          const int gm = cm0 - H + rho;
          const bool in_utt = gm >= clo && gm < chi;
          f32x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = (a[e] + b[e]) + xr[s][e];
          xr[s] = v;
          if (!last) {
            f32x4 w;
#pragma unroll
            for (int e = 0; e < 4; ++e) w[e] = in_utt ? (v[e] > 0.f ? v[e] : v[e] * slope) : 0.f;
            *reinterpret_cast<f32x4*>(sX + rho * LDA + n_e) = w;
          } else if (in_utt && rho >= H && rho < RX - H) {
            if (p.R2) {
              const f32x4 rr = *reinterpret_cast<const f32x4*>(p.R2 + (size_t)gm * p.ldr2 + n_e);
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = rr[e] + v[e];
            }
            if (p.div > 0.f) {
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = v[e] / p.div;
            }
            *reinterpret_cast<f32x4*>(p.Y + (size_t)gm * p.ldy + n_e) = v;
          }
        };
        if (s0 == 0 && s1 == SLOTS) rb_conv<C, TAPS, 0, SLOTS>(sM, sW, 1, tw, jw, r, g, epi);
        else if (s0 == 0) rb_conv<C, TAPS, 0, SLOTS - 1>(sM, sW, 1, tw, jw, r, g, epi);
        else if (s1 == SLOTS) rb_conv<C, TAPS, 1, SLOTS>(sM, sW, 1, tw, jw, r, g, epi);
        else rb_conv<C, TAPS, 1, SLOTS - 1>(sM, sW, 1, tw, jw, r, g, epi);
        vlo = lo; vhi = hi;
      }
      // new x slab complete, next conv's matrix in place (after the last pair the staging barrier makes it visible)
      if (pair < 2) conv_done(2 * pair + 2);
      else {
        if constexpr (!WDB) __syncthreads();
        w_commit(0);
      }
    }
  }
#endif
}

bool resblock_fused_eligible(int C, int taps, const int* dil, int ldx, int ldy, int nseg, long long M) {
  if (!((C == 16 || C == 32) && (taps == 3 || taps == 7 || taps == 11))) return false;
  const int H = (taps - 1) / 2 * (dil[0] + dil[1] + dil[2] + 3);
  // the valid range may lose at most the first and the last tile slot of a wave: H <= 64 rows per side
  return dil[0] >= 1 && dil[1] >= 1 && dil[2] >= 1 && H <= 64 && ldx == C && (ldy & 3) == 0 && nseg <= RB_MAXSEG && M > 0 &&
         slab_rows_ok(M);
}

static int rb_cus(int& cus) {             // CU count of the CURRENT device, read once per device (thread-safe)
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

template <int C, int TAPS>
static int launch_rb_t(const ResblockArgs& a, hipStream_t stream) {
  using G = RbGeom<C, TAPS>;
  static_assert(G::LDS_BYTES <= 160 * 1024, "slabs + one weight matrix must fit the CU's LDS");
  SS_MAX_LDS_ONCE((&resblock_fused_kernel<C, TAPS>), G::LDS_BYTES);
  int cus = 0;
  { int rc = rb_cus(cus); if (rc != SS_OK) return rc; }
  const int H = G::HT * (a.dil[0] + a.dil[1] + a.dil[2] + 3);
  const int BM = G::RX - 2 * H;
  const int nseg = a.nseg > 0 ? a.nseg : 1;
  const long long max_blocks = (long long)cdiv(a.M, BM) + nseg;            // upper bound (per-utterance round-up)
  const int grid = (int)std::min<long long>(cus, std::max<long long>(1, max_blocks));   // one workgroup per CU (LDS)
  // profiler classes resblock_fused<32> / <16> (20 / 21); algorithmic work of the six convs: 12 M C^2 k FLOP; bytes: x read
  // once, y written once, the MRF accumulator read once when present, the six weight matrices + biases once
  GemmArgs ga;
  ga.M = a.M; ga.N = C; ga.Cin = C; ga.taps = TAPS; ga.in_len = a.M; ga.R2 = a.R2;
  ga.algo_flops = 12.0 * (double)a.M * C * C * TAPS;
  ga.algo_bytes = 4.0 * ((double)a.M * C * (2 + (a.R2 ? 1 : 0)) + 6.0 * C * (C * TAPS + 1));
  ProfRec rec{}; bool prof = false;
  int rc = prof_begin(ga, stream, C == 32 ? 20 : 21, rec, prof);
  if (rc != SS_OK) return rc;
  hipLaunchKernelGGL((resblock_fused_kernel<C, TAPS>), dim3(grid), dim3(RB_NW * 64), G::LDS_BYTES, stream, a);
  SS_LAUNCH_CHECK();
  return prof_end(stream, rec, prof);
}

int launch_resblock_fused(const float* X, int ldx, const float* const* W1, const float* const* B1, const float* const* W2,
                          const float* const* B2, const int* dil, float* Y, int ldy, const float* R2, int ldr2, float div, int C,
                          int taps, int M, float slope, const int* segs, int nseg, hipStream_t stream) {
  if (!resblock_fused_eligible(C, taps, dil, ldx, ldy, nseg, M) || X == Y) return SS_ERR_ARG;
  ResblockArgs a;
  a.X = X; a.ldx = ldx; a.Y = Y; a.ldy = ldy; a.R2 = R2; a.ldr2 = ldr2; a.div = div; a.M = M; a.slope = slope;
  a.segs = segs; a.nseg = nseg;
  for (int i = 0; i < 3; ++i) { a.W1[i] = W1[i]; a.B1[i] = B1[i]; a.W2[i] = W2[i]; a.B2[i] = B2[i]; a.dil[i] = dil[i]; }
#define SS_RB(C_, T_) if (C == C_ && taps == T_) return launch_rb_t<C_, T_>(a, stream);
  SS_RB(32, 3) SS_RB(32, 7) SS_RB(32, 11) SS_RB(16, 3) SS_RB(16, 7) SS_RB(16, 11)
#undef SS_RB
  return SS_ERR_ARG;
}

}  // namespace ss
