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

template <int C, int TAPS> struct RbGeom {
  static constexpr int TN = C / 16, LDA = C + 4, K = TAPS * C, LDW = K + 4, HT = (TAPS - 1) / 2;
  // slab tiles: bounded by 160 KB of LDS (two slabs with one margin tile on either side + one weight matrix)
  static constexpr int NTILE = C == 32 ? (TAPS >= 11 ? 23 : 24) : 40;
  static constexpr int SLOTS = (NTILE + RB_NW - 1) / RB_NW;
  static constexpr int RX = NTILE * 16;
  static constexpr int W_FLOATS = (C * LDW + 255) & ~255;
  static constexpr int SLAB_FLOATS = ((NTILE + 2) * 16 * LDA + 255) & ~255;
  static constexpr size_t LDS_BYTES = (size_t)(W_FLOATS + 2 * SLAB_FLOATS) * sizeof(float) + (RB_MAXSEG + 2) * sizeof(int);
  static constexpr int WV4 = (C * (K / 4) + RB_NW * 64 - 1) / (RB_NW * 64);   // weight float4 per thread
};

// One conv of the ResBlock for the wave's tile slots [S0, S1): acc = sum_tap sum_cc sum_e  W . src^T out of LDS.
// `drows` = rows per tap (the dilation), the tap-0 row of output row rho is rho - HT * drows.
template <int C, int TAPS, int S0, int S1, class Epi>
__device__ __forceinline__ void rb_conv(const float* __restrict__ src, const float* __restrict__ sW, int drows, int wave,
                                        int r, int g, Epi&& epi) {
  using G = RbGeom<C, TAPS>;
  constexpr int TN = G::TN, LDA = G::LDA, LDW = G::LDW, NS = S1 - S0;
  if constexpr (NS > 0) {
    f32x4 acc[NS][TN];
#pragma unroll
    for (int u = 0; u < NS; ++u)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[u][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* pa = src + ((S0 * RB_NW + wave) * 16 + r - G::HT * drows) * LDA + 4 * g;
    const float* pw = sW + r * LDW + 4 * g;
    const int a_step = drows * LDA;
    for (int tap = 0; tap < TAPS; ++tap) {
#pragma unroll
      for (int cc = 0; cc < C / 16; ++cc) {
        f32x4 af[NS], bf[TN];
#pragma unroll
        for (int u = 0; u < NS; ++u) af[u] = *reinterpret_cast<const f32x4*>(pa + u * RB_NW * 16 * LDA + cc * 16);
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const f32x4*>(pw + j * 16 * LDW + cc * 16);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int u = 0; u < NS; ++u)
#pragma unroll
            for (int j = 0; j < TN; ++j)
              acc[u][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[j][e], af[u][e], acc[u][j], 0, 0, 0);   // D = W . A^T
      }
      pa += a_step;
      pw += C;
    }
#pragma unroll
    for (int u = 0; u < NS; ++u)
#pragma unroll
      for (int j = 0; j < TN; ++j) epi(S0 + u, j, acc[u][j]);
  }
}

template <int C, int TAPS>
__global__ __launch_bounds__(RB_NW * 64) void resblock_fused_kernel(const ResblockArgs p) {
#if __HIP_DEVICE_COMPILE__
  using G = RbGeom<C, TAPS>;
  constexpr int TN = G::TN, LDA = G::LDA, K = G::K, LDW = G::LDW, HT = G::HT, NTILE = G::NTILE, SLOTS = G::SLOTS, RX = G::RX;
  constexpr int NT = RB_NW * 64, WV4 = G::WV4;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sW = smem;                                          // [C][LDW]
  float* sX = sW + G::W_FLOATS + 16 * LDA;                   // activated x slab, row 0 after one margin tile
  float* sM = sW + G::W_FLOATS + G::SLAB_FLOATS + 16 * LDA;  // mid slab
  int* s_blk = reinterpret_cast<int*>(sW + G::W_FLOATS + 2 * G::SLAB_FLOATS);

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int r = lane & 15, g = lane >> 4;
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
  for (int i = t; i < 2 * G::SLAB_FLOATS; i += NT) sW[G::W_FLOATS + i] = 0.f;
  __syncthreads();
  const int nblocks = s_blk[nseg];

  // weights of conv `c` (0..5: pair c/2, conv1 / conv2) global -> registers -> LDS
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
  auto w_commit = [&]() {
#pragma unroll
    for (int u = 0; u < WV4; ++u) {
      const int idx = t + u * NT;
      if (idx < C * (K / 4)) {
        const int n = idx / (K / 4), k4 = idx - n * (K / 4);
        *reinterpret_cast<f32x4*>(sW + n * LDW + k4 * 4) = wreg[u];
      }
    }
  };

  // block geometry + the raw x rows of this wave's tiles (epilogue layout: lane (r, g) holds row tau*16 + r,
  // channels j*16 + 4g .. +3)
  int seg = 0, seg_lo = 0, seg_hi = 0, m0 = 0;
  auto locate = [&](int blk) {                 // blocks ascend per workgroup
    while (blk >= s_blk[seg + 1]) ++seg;
    seg_lo = p.nseg > 0 ? p.segs[4 * seg] : 0;
    seg_hi = seg_lo + (p.nseg > 0 ? p.segs[4 * seg + 1] : p.M);
    m0 = seg_lo + (blk - s_blk[seg]) * BM;
  };
  f32x4 pre[SLOTS][TN], xr[SLOTS][TN];
  auto x_fetch = [&]() {
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
      const int tau = s * RB_NW + wave;
      const int gm = m0 - H + tau * 16 + r;
      const bool ok = tau < NTILE && gm >= seg_lo && gm < seg_hi;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (ok) v = *reinterpret_cast<const f32x4*>(p.X + (size_t)gm * p.ldx + j * 16 + 4 * g);
        pre[s][j] = v;
      }
    }
  };

  int blk = blockIdx.x;
  if (blk < nblocks) {
    locate(blk);
    x_fetch();
    w_fetch(0);
    w_commit();
  }
  for (; blk < nblocks; blk += gridDim.x) {
    const int cm0 = m0, clo = seg_lo, chi = seg_hi;
    // ---- stage: raw x -> registers, lrelu(x) -> slab (rows outside the utterance were fetched as zeros) ----
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
      const int tau = s * RB_NW + wave;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const f32x4 v = pre[s][j];
        xr[s][j] = v;
        if (tau < NTILE) {
          f32x4 a;
#pragma unroll
          for (int e = 0; e < 4; ++e) a[e] = v[e] > 0.f ? v[e] : v[e] * slope;
          *reinterpret_cast<f32x4*>(sX + (tau * 16 + r) * LDA + j * 16 + 4 * g) = a;
        }
      }
    }
    const bool more = blk + (int)gridDim.x < nblocks;
    __syncthreads();                                       // slab + weights of conv 0 visible

    int vlo = 0, vhi = RX;                                 // valid rows of the x slab entering the pair
#pragma unroll 1
    for (int pair = 0; pair < 3; ++pair) {
      const int d = p.dil[pair];
      const int h1 = HT * d, h2 = HT;
      // ================= conv1 (dilated): activated x slab -> mid slab =================
      {
        const int lo = vlo + h1, hi = vhi - h1;            // valid mid rows
        const int s0 = (wave * 16 + 16 > lo) ? 0 : 1;      // slot 0 of this wave intersects the range?
        const int tl = (SLOTS - 1) * RB_NW + wave;         // last slot's tile
        const int s1 = (tl < NTILE && tl * 16 < hi) ? SLOTS : SLOTS - 1;
        w_fetch(2 * pair + 1);                             // conv2's matrix, under the MFMAs
        const float* b1 = p.B1[pair];
        auto epi = [&](int s, int j, const f32x4& a) {
          const int rho = (s * RB_NW + wave) * 16 + r;
          const int gm = cm0 - H + rho;
          const bool in_utt = gm >= clo && gm < chi;
          const f32x4 b = *reinterpret_cast<const f32x4*>(b1 + j * 16 + 4 * g);
          f32x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float x = a[e] + b[e];
            v[e] = in_utt ? (x > 0.f ? x : x * slope) : 0.f;
          }
          *reinterpret_cast<f32x4*>(sM + rho * LDA + j * 16 + 4 * g) = v;
        };
        if (s0 == 0 && s1 == SLOTS) rb_conv<C, TAPS, 0, SLOTS>(sX, sW, d, wave, r, g, epi);
        else if (s0 == 0) rb_conv<C, TAPS, 0, SLOTS - 1>(sX, sW, d, wave, r, g, epi);
        else if (s1 == SLOTS) rb_conv<C, TAPS, 1, SLOTS>(sX, sW, d, wave, r, g, epi);
        else rb_conv<C, TAPS, 1, SLOTS - 1>(sX, sW, d, wave, r, g, epi);
        vlo = lo; vhi = hi;
      }
      __syncthreads();                                     // mid complete, everyone is done with conv1's matrix
      w_commit();
      __syncthreads();
      // ================= conv2 (dilation 1): mid slab -> x (+ residual) =================
      {
        const int lo = vlo + h2, hi = vhi - h2;            // valid rows of the new x
        const int s0 = (wave * 16 + 16 > lo) ? 0 : 1;
        const int tl = (SLOTS - 1) * RB_NW + wave;
        const int s1 = (tl < NTILE && tl * 16 < hi) ? SLOTS : SLOTS - 1;
        const bool last = pair == 2;
        if (!last) w_fetch(2 * pair + 2);                  // next pair's conv1
        else {
          w_fetch(0);                                      // next block starts with conv 0 again
          if (more) { locate(blk + gridDim.x); x_fetch(); }
        }
        const float* b2 = p.B2[pair];
        auto epi = [&](int s, int j, const f32x4& a) {
          const int rho = (s * RB_NW + wave) * 16 + r;
          const int gm = cm0 - H + rho;
          const bool in_utt = gm >= clo && gm < chi;
          const int n = j * 16 + 4 * g;
          const f32x4 b = *reinterpret_cast<const f32x4*>(b2 + n);
          f32x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = (a[e] + b[e]) + xr[s][j][e];
          xr[s][j] = v;
          if (!last) {
            f32x4 w;
#pragma unroll
            for (int e = 0; e < 4; ++e) w[e] = in_utt ? (v[e] > 0.f ? v[e] : v[e] * slope) : 0.f;
            *reinterpret_cast<f32x4*>(sX + rho * LDA + n) = w;
          } else if (in_utt && rho >= H && rho < RX - H) {
            if (p.R2) {
              const f32x4 rr = *reinterpret_cast<const f32x4*>(p.R2 + (size_t)gm * p.ldr2 + n);
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = rr[e] + v[e];
            }
            if (p.div > 0.f) {
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = v[e] / p.div;
            }
            *reinterpret_cast<f32x4*>(p.Y + (size_t)gm * p.ldy + n) = v;
          }
        };
        if (s0 == 0 && s1 == SLOTS) rb_conv<C, TAPS, 0, SLOTS>(sM, sW, 1, wave, r, g, epi);
        else if (s0 == 0) rb_conv<C, TAPS, 0, SLOTS - 1>(sM, sW, 1, wave, r, g, epi);
        else if (s1 == SLOTS) rb_conv<C, TAPS, 1, SLOTS>(sM, sW, 1, wave, r, g, epi);
        else rb_conv<C, TAPS, 1, SLOTS - 1>(sM, sW, 1, wave, r, g, epi);
        vlo = lo; vhi = hi;
      }
      __syncthreads();                                     // new x slab complete, everyone is done with conv2's matrix
      w_commit();
      if (!last_pair_dummy(pair)) __syncthreads();
    }
  }
#endif
}

}  // namespace ss
