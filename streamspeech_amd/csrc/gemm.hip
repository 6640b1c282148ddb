// Implicit-GEMM Conv1d / Linear kernel for gfx950, exact-f32 MFMA (see gemm.hpp).
//
// Tiling: a 256-thread workgroup (4 wave64) owns a BM x BN output tile; the K loop walks
// (tap, channel-block) pairs in BK-wide steps.  Per step the A rows (shifted by the tap) and the
// W rows are staged global -> registers -> LDS (register prefetch of step k+1 overlaps the MFMAs
// of step k, one barrier per step), then each wave feeds v_mfma_f32_16x16x4_f32 from LDS with
// ds_read_b128: lane (r = lane&15, g = lane>>4) reads 4 consecutive k of row r at k-offset 4g
// and issues 4 MFMAs; MFMA i therefore contracts k = {i, 4+i, 8+i, 12+i} of the 16-wide slab --
// a permutation of k shared by A and B, which leaves the sum unchanged.
// C/D layout (MI355X guide §3): col = lane&15, row = (lane>>4)*4 + reg.
#include "gemm.hpp"

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <mutex>
#include <tuple>
#include <type_traits>
#include <utility>
#include <vector>

namespace ss {

using f32x4 = __attribute__((ext_vector_type(4))) float;

// KS > 1: intra-workgroup split-K.  The workgroup has KS groups of 4 waves; every group owns the
// whole BM x BN tile but only 1/KS of the k-steps, with private LDS staging, and the partial tiles
// are summed (fixed order) through LDS before the epilogue.  This fills all 1024 SIMDs when M*N
// alone gives too few tiles (early vocoder stages, M~500 decoder GEMMs) without atomics or a
// second launch.
//
// BLK (the pack-invariant CANON_SEQ form, gemm.hpp): the k walk is cut into blocks of CANON_KBLOCK = 64 -- one accumulator chain inside a
// block, the block sums added in ascending order into a second accumulator set ("tot += acc; acc = 0" every 64 / BK steps).  A
// function of the walk alone (taps, Cin), not of M or the grid.  Rounding of a K-term sum then grows like a 64-term chain + K / 64
// adds instead of a K-term chain (the K = 256 chains sat 1.35-1.5x, the K = 2048 / 5120 ones 2.8-3.8x farther from
// float64 than torch's CPU sgemm on the GPU box; this form: ~0.7x).
template <int BM, int BN, int BK, int WM, int WN, int KS, int PD, bool BLK = false>
__global__ __launch_bounds__(256 * KS) void conv_gemm_kernel(const GemmArgs p) {
  static_assert(WM * WN == 4, "4 waves per k-group");
  static_assert(!BLK || KS == 1, "the blocked chain is the KS = 1 form");
  // row stride in floats: 16-B aligned and = 10 (BK=32) / 6 (BK=16) sixteen-byte slots, which makes the
  // 16-lane groups of ds_read_b128 hit 16 distinct slots (conflict-free; +4 was 2-way)
  constexpr int LDK = BK + 8;
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int TM = WTM / 16, TN = WTN / 16;
  static_assert(TM >= 1 && TN >= 1, "wave tile must hold a 16x16 MFMA tile");
  constexpr int KQ = BK / 4;
  constexpr int A_F4 = BM * KQ, W_F4 = BN * KQ;
  constexpr int NA = (A_F4 + 255) / 256, NW = (W_F4 + 255) / 256;
  constexpr int STAGE = (BM + BN) * LDK;
  extern __shared__ __attribute__((aligned(16))) float smem_all[];
  const int kg = threadIdx.x >> 8;                     // k-group of this thread
  float* smem = smem_all + kg * 2 * STAGE;

  const int t = threadIdx.x & 255, lane = t & 63, wave = t >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int r = lane & 15, g = lane >> 4;

  int out_start = 0, out_len = p.M, in_start = 0, in_len = p.in_len;
  if (p.nseg > 0) {
    const int* s = p.segs + 4 * blockIdx.z;
    out_start = s[0]; out_len = s[1]; in_start = s[2]; in_len = s[3];
  }
  const int m0 = blockIdx.x * BM + p.m_begin;
  if (m0 >= out_len) return;
  const int n0 = blockIdx.y * BN;
  const int kpt = p.Cin / BK;          // k-steps per tap
  const int nk = p.taps * kpt;
  const int Ktot = p.taps * p.Cin;

  // per-thread staging slots
  int a_rin0[NA], a_lim[NA], a_c4[NA], a_lds[NA];
  bool a_ok[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int f = t + i * 256;
    const int row = f / KQ, c4 = f % KQ;
    const int m = m0 + row;
    a_ok[i] = (f < A_F4) && (m < out_len);
    a_rin0[i] = m * p.stride - p.pad;
    a_lim[i] = p.chunk > 0 ? ((m * p.stride) / p.chunk + 1) * p.chunk : 0x7fffffff;
    a_c4[i] = c4 * 4;
    a_lds[i] = row * LDK + c4 * 4;
  }
  int w_c4[NW], w_lds[NW];
  size_t w_off[NW];
  bool w_ok[NW];
#pragma unroll
  for (int i = 0; i < NW; ++i) {
    const int f = t + i * 256;
    const int row = f / KQ, c4 = f % KQ;
    const int n = n0 + row;
    w_ok[i] = (f < W_F4) && (n < p.N);
    w_off[i] = (size_t)n * Ktot + c4 * 4;
    w_c4[i] = c4 * 4;
    w_lds[i] = BM * LDK + row * LDK + c4 * 4;
  }

  // Register ring of PD k-steps: the global loads of step k+PD are issued before the MFMAs of
  // step k, so a load has PD steps of MFMA work to land before its turn to be written to LDS
  // (the leaky-ReLU input activation is applied at that write, never right after the load).
  f32x4 ra[PD][NA], rw[PD][NW];
  auto load_regs = [&](auto SLOT, bool valid, int kb, int tap, int ci0) {
    constexpr int sl = decltype(SLOT)::value;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int rin = a_rin0[i] + tap * p.dil;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (valid && a_ok[i] && rin >= 0 && rin < in_len && rin < a_lim[i])
        v = *reinterpret_cast<const f32x4*>(p.A + (size_t)(in_start + rin) * p.lda + ci0 + a_c4[i]);
      ra[sl][i] = v;
    }
#pragma unroll
    for (int i = 0; i < NW; ++i) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (valid && w_ok[i]) v = *reinterpret_cast<const f32x4*>(p.W + w_off[i] + (size_t)tap * p.Cin + ci0);
      rw[sl][i] = v;
    }
  };
  const bool lrelu = p.in_act == ACT_LRELU;
  const float slope = p.in_slope;
  auto store_lds = [&](auto SLOT, int buf) {
    constexpr int sl = decltype(SLOT)::value;
    float* base = smem + buf * STAGE;
#pragma unroll
    for (int i = 0; i < NA; ++i)
      if (t + i * 256 < A_F4) {
        f32x4 v = ra[sl][i];
        if (lrelu) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * slope;
        }
        *reinterpret_cast<f32x4*>(base + a_lds[i]) = v;
      }
#pragma unroll
    for (int i = 0; i < NW; ++i)
      if (t + i * 256 < W_F4) *reinterpret_cast<f32x4*>(base + w_lds[i]) = rw[sl][i];
  };

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = {0.f, 0.f, 0.f, 0.f};
  [[maybe_unused]] f32x4 tot[BLK ? TM : 1][BLK ? TN : 1];
  [[maybe_unused]] bool flushed = false;
  constexpr int BLK_STEPS = CANON_KBLOCK / BK;

  // k-steps of this group: [kb0, kb1); every group runs `kper` iterations (steps past kb1 load
  // zeros) so barriers and the MFMA body stay uniform
  const int kper = (nk + KS - 1) / KS;
  const int kb0 = kg * kper;
  const int kb1 = min(nk, kb0 + kper);
  // k-step s = (channel block s / taps, tap s % taps): the taps of one channel block run back to back
  // so the tile's A rows are re-read while still in L1/L2 (see conv_sk.hip)
  int nci = (kb0 / p.taps) * BK;             // (tap, channel offset) of the next step to load
  int ntap = kb0 - (kb0 / p.taps) * p.taps;
  auto advance = [&]() { if (++ntap >= p.taps) { ntap = 0; nci += BK; } };
  auto for_slots = [&](auto&& fn) {          // fn(integral_constant<u>) for u = 0..PD-1
    fn(std::integral_constant<int, 0>{});
    if constexpr (PD > 1) fn(std::integral_constant<int, 1>{});
    if constexpr (PD > 2) fn(std::integral_constant<int, 2>{});
    if constexpr (PD > 3) fn(std::integral_constant<int, 3>{});
  };
  for_slots([&](auto U) {
    constexpr int u = decltype(U)::value;
    load_regs(U, kb0 + u < kb1, kb0 + u, ntap, nci);
    advance();
  });
  store_lds(std::integral_constant<int, 0>{}, 0);
  __syncthreads();

  for (int it0 = 0; it0 < kper; it0 += PD) {
    for_slots([&](auto U) {
      constexpr int u = decltype(U)::value;
      const int it = it0 + u;
      if (it < kper) {                       // block-uniform
        const int cur = it & 1;
        load_regs(U, kb0 + it + PD < kb1, kb0 + it + PD, ntap, nci);   // slot u (step it) is already in LDS
        advance();
        const float* As = smem + cur * STAGE + (wm * WTM + r) * LDK + g * 4;
        const float* Ws = smem + cur * STAGE + BM * LDK + (wn * WTN + r) * LDK + g * 4;
#pragma unroll
        for (int kk = 0; kk < BK / 16; ++kk) {
          f32x4 af[TM], bf[TN];
#pragma unroll
          for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const f32x4*>(As + i * 16 * LDK + kk * 16);
#pragma unroll
          for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const f32x4*>(Ws + j * 16 * LDK + kk * 16);
#pragma unroll
          for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
              for (int j = 0; j < TN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[j][e], af[i][e], acc[i][j], 0, 0, 0);   // D = W.A^T (see epilogue)
        }
        if constexpr (BLK) {
          if ((it + 1) % BLK_STEPS == 0 && it + 1 < kper) {      // block boundary of the walk (block-uniform)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
              for (int j = 0; j < TN; ++j) {
#pragma unroll
                for (int e = 0; e < 4; ++e) tot[i][j][e] = flushed ? tot[i][j][e] + acc[i][j][e] : acc[i][j][e];
                acc[i][j] = {0.f, 0.f, 0.f, 0.f};
              }
            flushed = true;
          }
        }
        if (it + 1 < kper) store_lds(std::integral_constant<int, (u + 1) % PD>{}, cur ^ 1);
        __syncthreads();
      }
    });
  }
  if constexpr (BLK) {
    if (flushed) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[i][j][e] = tot[i][j][e] + acc[i][j][e];
    }
  }

  if (KS > 1) {
    // partial tiles of groups 1..KS-1 -> LDS (staging buffers are dead now), group 0 sums in order
    f32x4* red = reinterpret_cast<f32x4*>(smem_all);
    constexpr int PER_GROUP = 4 * TM * TN * 64;
    if (kg > 0) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          red[(kg - 1) * PER_GROUP + ((wave * TM + i) * TN + j) * 64 + lane] = acc[i][j];
    }
    __syncthreads();
    if (kg > 0) return;
#pragma unroll
    for (int s2 = 1; s2 < KS; ++s2)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const f32x4 o = red[(s2 - 1) * PER_GROUP + ((wave * TM + i) * TN + j) * 64 + lane];
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[i][j][e] += o[e];
        }
  }

  // ---- epilogue ----
  auto activate = [&](float v) -> float {
    switch (p.act) {
      case ACT_SILU: return v / (1.0f + expf(-v));
      case ACT_RELU: return fmaxf(v, 0.f);
      case ACT_TANH: return tanhf(v);
      case ACT_LRELU: return v > 0.f ? v : v * p.act_slope;
      default: return v;
    }
  };
  // The MFMAs are issued with the operands swapped (D = W_tile . A_tile^T): in the C/D layout
  // (col = lane&15, row = 4*(lane>>4) + reg) a lane then holds 4 CONSECUTIVE output channels
  // n = 4g..4g+3 of ONE row m = r, so bias / residual / output move as float4 whenever the leading
  // dimensions allow it (4x fewer memory instructions; scalar fallback otherwise, e.g. N = 1005).
  if (p.glu) {
    if constexpr (TN % 2 == 0) {
      const bool vec = (p.ldc & 3) == 0;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int m = m0 + wm * WTM + i * 16 + r;
        if (m >= out_len) continue;
        float* crow = p.C + (size_t)(out_start + m) * p.ldc;
#pragma unroll
        for (int j = 0; j < TN; j += 2) {
          const int ncol = n0 + wn * WTN + j * 16;      // start of the [value|gate] 32-col block
          if (ncol + 31 >= p.N) continue;
          const int oc = ncol / 2 + g * 4;              // first of this lane's 4 output channels
          f32x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float bv = p.bias ? p.bias[ncol + g * 4 + e] : 0.f;
            const float bg = p.bias ? p.bias[ncol + 16 + g * 4 + e] : 0.f;
            const float val = acc[i][j][e] + bv;
            const float gate = acc[i][j + 1][e] + bg;
            o[e] = val * (1.0f / (1.0f + expf(-gate)));
          }
          if (vec) *reinterpret_cast<f32x4*>(crow + oc) = o;
          else {
#pragma unroll
            for (int e = 0; e < 4; ++e) crow[oc + e] = o[e];
          }
        }
      }
    }
    return;
  }
  const bool vec = ((p.ldc | (p.R ? p.ldr : 0) | (p.R2 ? p.ldr2 : 0) | (p.C2 ? p.ldc2 : 0) | p.N) & 3) == 0;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int m = m0 + wm * WTM + i * 16 + r;
    if (m >= out_len) continue;
    const size_t row = (size_t)(out_start + m);
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + wn * WTN + j * 16 + g * 4;
      if (n >= p.N) continue;
      f32x4 v = acc[i][j];
      if (vec) {
        if (p.bias) {
          const f32x4 b = *reinterpret_cast<const f32x4*>(p.bias + n);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += b[e];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = activate(v[e]) * p.alpha;
        if (p.R) {
          const f32x4 rr = *reinterpret_cast<const f32x4*>(p.R + row * p.ldr + n);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += rr[e];
        }
        if (p.R2) {
          const f32x4 rr = *reinterpret_cast<const f32x4*>(p.R2 + row * p.ldr2 + n);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = rr[e] + v[e];
        }
        if (p.div > 0.f) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = v[e] / p.div;
        }
        *reinterpret_cast<f32x4*>(p.C + row * p.ldc + n) = v;
        if (p.C2) {
          f32x4 w2;
#pragma unroll
          for (int e = 0; e < 4; ++e) w2[e] = v[e] > 0.f ? v[e] : v[e] * p.c2_slope;
          *reinterpret_cast<f32x4*>(p.C2 + row * p.ldc2 + n) = w2;
        }
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (n + e >= p.N) continue;
          float x = activate(v[e] + (p.bias ? p.bias[n + e] : 0.f)) * p.alpha;
          if (p.R) x += p.R[row * p.ldr + n + e];
          if (p.R2) x = p.R2[row * p.ldr2 + n + e] + x;
          if (p.div > 0.f) x = x / p.div;
          p.C[row * p.ldc + n + e] = x;
          if (p.C2) p.C2[row * p.ldc2 + n + e] = x > 0.f ? x : x * p.c2_slope;
        }
      }
    }
  }
}


// =================================================================================================
// Small-M linear layers (M <= 128: every encoder/decoder projection at batch 1, decode steps at
// M = 1).  These are latency-bound, so there is no LDS staging and no barrier in the main loop:
// a workgroup owns 16 rows x (16*WN) columns, its 4 waves are arranged SK (split-K) x WN, and
// each wave feeds v_mfma_f32_16x16x4_f32 straight from global memory -- lane (r, g) loads the
// float4 at [row r][k + 4g] of A and of W (64-B row segments), 4 MFMAs per pair of loads, loads
// of 4 chunks kept in flight.  The SK partial tiles are summed through LDS once at the end.
// Optional fused LayerNorm over K on the A rows (two-pass statistics per wave, rows stay in L1).
// =================================================================================================
template <int SK, int WN, bool LNORM, bool GLU = false>
__global__ __launch_bounds__(256) void smallm_gemm_kernel(const GemmArgs p) {
  static_assert(SK * WN == 4, "4 waves");
  static_assert(!GLU || (SK == 2 && WN == 2), "GLU: wave column 0 = the 16 value columns, 1 = the 16 gate columns of a 32-column block");
  __shared__ f32x4 red[SK > 1 ? (SK - 1) * WN * 64 : 1];
  __shared__ float ln_stat[32];                   // mean[16], rstd[16] of the workgroup's rows
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int sk = wave / WN, wn = wave % WN;
  const int r = lane & 15, g = lane >> 4;
  // n-tiles run fastest and their count is a multiple of 8 (launch_smallm): blocks land on XCD (linear id % 8), so the W rows
  // of an n-tile are fetched by ONE XCD's L2 for all m-tiles instead of by as many XCDs as there are m-tiles (the rows are
  // the small operand here; MI355X_MICROARCH.md "block b runs on XCD b % 8" -- a speed affinity, nothing depends on it)
  if ((int)blockIdx.x * (16 * WN) >= p.N) return;
  const int m0 = blockIdx.y * 16;
  const int n0 = blockIdx.x * (16 * WN) + wn * 16;
  const int K = p.Cin;
  const int KC = K / 16;                          // 16-wide chunks
  const int c_begin = (KC * sk) / SK, c_end = (KC * (sk + 1)) / SK;

  const int am = m0 + r;
  const bool a_ok = am < p.M;
  const float* arow = p.A + (size_t)(a_ok ? am : 0) * p.lda + 4 * g;
  const int wnr = n0 + r;
  const bool w_ok = wnr < p.N;
  const float* wrow = p.W + (size_t)(w_ok ? wnr : 0) * K + 4 * g;

  float mean = 0.f, rstd = 1.f;
  if (LNORM) {
    // 16 threads per row, each keeps its K/16 values in registers (K <= 512): one read, two-pass stats
    const int lr = t >> 4, part = t & 15;
    const bool ok = (m0 + lr) < p.M;
    const float* xr = p.A + (size_t)(ok ? m0 + lr : 0) * p.lda + part * 4;
    f32x4 v[8];
    float sm = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      v[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (j * 64 < K && ok) v[j] = *reinterpret_cast<const f32x4*>(xr + j * 64);
      sm += (v[j][0] + v[j][1]) + (v[j][2] + v[j][3]);
    }
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) sm += __shfl_xor(sm, o, 64);
    const float mu = sm / (float)K;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (j * 64 < K) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d = v[j][e] - mu; q += d * d; }
      }
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) q += __shfl_xor(q, o, 64);
    if (part == 0) { ln_stat[lr] = mu; ln_stat[16 + lr] = 1.0f / sqrtf(q / (float)K + 1e-5f); }
    __syncthreads();
    mean = ln_stat[r];
    rstd = ln_stat[16 + r];
  }

  f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};   // 2 chains: MFMA dependent latency 40 > issue 32
  f32x4 tot = {0.f, 0.f, 0.f, 0.f};
  // a wave whose k-range is at most 128 (K <= 512 on four waves) already runs two chains of <= 64 terms: nothing to cut.  A function of
  // (K, SK) alone -- never of M -- so a row's bits stay the same in any pack.
  const bool blocked = (c_end - c_begin) > 8;
  auto load4 = [&](f32x4 (&a)[4], f32x4 (&w)[4], int c) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      a[u] = f32x4{0.f, 0.f, 0.f, 0.f};
      w[u] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (a_ok) a[u] = *reinterpret_cast<const f32x4*>(arow + (c + u) * 16);
      if (w_ok) w[u] = *reinterpret_cast<const f32x4*>(wrow + (c + u) * 16);
    }
  };
  auto mma = [&](f32x4 a, f32x4 w, int c) {
    if (LNORM) {
      const f32x4 gm = *reinterpret_cast<const f32x4*>(p.ln_g + c * 16 + 4 * g);
      const f32x4 bt = *reinterpret_cast<const f32x4*>(p.ln_b + c * 16 + 4 * g);
#pragma unroll
      for (int e = 0; e < 4; ++e) a[e] = a_ok ? (a[e] - mean) * rstd * gm[e] + bt[e] : 0.f;
    }
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], w[0], acc, 0, 0, 0);
    acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], w[1], acc2, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], w[2], acc, 0, 0, 0);
    acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], w[3], acc2, 0, 0, 0);
    if (blocked && (c & 3) == 3) {  // end of a 64-wide k-block (round 6, gemm.hpp CANON_KBLOCK): block sums added in ascending order
#pragma unroll
      for (int e = 0; e < 4; ++e) { tot[e] += acc[e] + acc2[e]; acc[e] = 0.f; acc2[e] = 0.f; }
    }
  };
  auto mma4 = [&](f32x4 (&a)[4], f32x4 (&w)[4], int c) {
#pragma unroll
    for (int u = 0; u < 4; ++u) mma(a[u], w[u], c + u);
  };
  // groups of 4 chunks, register double-buffered: the loads of group g+1 fly under the MFMAs of g
  const int G = (c_end - c_begin) / 4;
  f32x4 a0[4], w0[4], a1[4], w1[4];
  if (G > 0) load4(a0, w0, c_begin);
  for (int gi = 0; gi < G; gi += 2) {
    const int c = c_begin + 4 * gi;
    if (gi + 1 < G) load4(a1, w1, c + 4);
    mma4(a0, w0, c);
    if (gi + 1 < G) {
      if (gi + 2 < G) load4(a0, w0, c + 8);
      mma4(a1, w1, c + 4);
    }
  }
  for (int c = c_begin + 4 * G; c < c_end; ++c) {
    f32x4 a = {0.f, 0.f, 0.f, 0.f}, w = {0.f, 0.f, 0.f, 0.f};
    if (a_ok) a = *reinterpret_cast<const f32x4*>(arow + c * 16);
    if (w_ok) w = *reinterpret_cast<const f32x4*>(wrow + c * 16);
    mma(a, w, c);
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) acc[e] = tot[e] + (acc[e] + acc2[e]);

  if (SK > 1) {
    if (sk > 0) red[((sk - 1) * WN + wn) * 64 + lane] = acc;
    __syncthreads();
    if (!GLU && sk > 0) return;
    if (sk == 0) {
#pragma unroll
      for (int s2 = 1; s2 < SK; ++s2) {
        const f32x4 o = red[((s2 - 1) * WN + wn) * 64 + lane];
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] += o[e];
      }
    }
  }
  if constexpr (GLU) {
    // out[m, 16 b + r] = (value + bias) * sigmoid(gate + bias): the gate wave hands its tile to the value wave through LDS
    // (same formula and packing as the tile kernel's GLU epilogue: [16 value | 16 gate] column blocks)
    __shared__ f32x4 gate_tile[64];
    const int ncol = n0 + r;                               // this lane's column of the packed [value | gate] matrix
    const float bcol = (p.bias && ncol < p.N) ? p.bias[ncol] : 0.f;
    if (sk == 0 && wn == 1) {
      f32x4 gt;
#pragma unroll
      for (int e = 0; e < 4; ++e) gt[e] = acc[e] + bcol;
      gate_tile[lane] = gt;
    }
    __syncthreads();
    if (sk != 0 || wn != 0 || ncol + 16 >= p.N) return;
    const f32x4 gt = gate_tile[lane];
    const int oc = (int)blockIdx.x * 16 + r;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int m = m0 + g * 4 + e;
      if (m < p.M) p.C[(size_t)m * p.ldc + oc] = (acc[e] + bcol) * (1.0f / (1.0f + expf(-gt[e])));
    }
    return;
  }
  const int n = n0 + r;
  if (n >= p.N) return;
  const float b = p.bias ? p.bias[n] : 0.f;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int m = m0 + g * 4 + e;
    if (m < p.M) {
      float v = acc[e] + b;
      switch (p.act) {
        case ACT_SILU: v = v / (1.0f + expf(-v)); break;
        case ACT_RELU: v = fmaxf(v, 0.f); break;
        default: break;
      }
      v *= p.alpha;
      if (p.R) v += p.R[(size_t)m * p.ldr + n];
      p.C[(size_t)m * p.ldc + n] = v;
    }
  }
}


// =================================================================================================
// GEMV form for decode steps (M <= 4 rows: one new MT token, B = 1): these stream 1-4 MB of weights
// for a few kFLOP, so the matrix cores are irrelevant and what counts is bytes in flight.  One wave
// per output column (KS waves per column when N alone gives too few waves to cover the chip): the
// lanes split K in float4 steps, so every weight row is read as fully coalesced 1-KB wave loads, all
// of a column's loads are issued before the first FMA, and the A rows (optionally LayerNorm'ed in
// registers: each wave holds the whole row) are loaded once per wave.  Wave-shuffle reduction, fused
// bias / activation / alpha / residual.  K % 256 == 0, K <= 2048.
// =================================================================================================
template <int MR, int KS, bool LNORM>
__global__ __launch_bounds__(256) void gemv_kernel(const GemmArgs p) {
  constexpr int WPB = 4;                       // waves per block
  constexpr int CPB = WPB / KS;                // columns per block
  __shared__ float part[WPB][MR];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int col = blockIdx.x * CPB + wave / KS;
  const int ks = wave % KS;
  const int K = p.Cin;
  const int kper = K / KS;                     // this wave's slice of K (multiple of 256)
  const int k0 = ks * kper;
  const int nit = kper / 256;                  // float4 steps per lane (<= 8)
  const bool col_ok = col < p.N;

  // weight slice of the column: issue everything first
  f32x4 wv[8];
  const float* wrow = p.W + (size_t)(col_ok ? col : 0) * K + k0 + 4 * lane;
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    wv[it] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (it < nit && col_ok) wv[it] = *reinterpret_cast<const f32x4*>(wrow + it * 256);
  }
  float acc[MR];
#pragma unroll
  for (int m = 0; m < MR; ++m) {
    acc[m] = 0.f;
    if (m >= p.M) continue;
    const float* xrow = p.A + (size_t)m * p.lda;
    float mean = 0.f, rstd = 1.f;
    if (LNORM) {                               // whole row per wave (K <= 2048: 8 float4 per lane), two-pass statistics
      f32x4 xv[8];
      float sm = 0.f;
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        xv[it] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (it * 256 < K) xv[it] = *reinterpret_cast<const f32x4*>(xrow + it * 256 + 4 * lane);
        sm += (xv[it][0] + xv[it][1]) + (xv[it][2] + xv[it][3]);
      }
      mean = wave_sum(sm) / (float)K;
      float q = 0.f;
#pragma unroll
      for (int it = 0; it < 8; ++it)
        if (it * 256 < K) {
#pragma unroll
          for (int e = 0; e < 4; ++e) { const float d = xv[it][e] - mean; q += d * d; }
        }
      rstd = 1.0f / sqrtf(wave_sum(q) / (float)K + 1e-5f);
    }
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
    for (int it = 0; it < 8; ++it)
      if (it < nit) {
        f32x4 x = *reinterpret_cast<const f32x4*>(xrow + k0 + it * 256 + 4 * lane);
        if (LNORM) {
          const f32x4 gm = *reinterpret_cast<const f32x4*>(p.ln_g + k0 + it * 256 + 4 * lane);
          const f32x4 bt = *reinterpret_cast<const f32x4*>(p.ln_b + k0 + it * 256 + 4 * lane);
#pragma unroll
          for (int e = 0; e < 4; ++e) x[e] = (x[e] - mean) * rstd * gm[e] + bt[e];
          // the waves of column 0 cover the whole row between them: they also publish the normalised row
          if (p.ln_out && blockIdx.x == 0 && wave < KS)
            *reinterpret_cast<f32x4*>(p.ln_out + (size_t)m * K + k0 + it * 256 + 4 * lane) = x;
        }
        a0 = fmaf(x[0], wv[it][0], a0); a1 = fmaf(x[1], wv[it][1], a1);
        a2 = fmaf(x[2], wv[it][2], a2); a3 = fmaf(x[3], wv[it][3], a3);
      }
    acc[m] = wave_sum((a0 + a1) + (a2 + a3));
  }
  if (KS > 1) {
    if (lane == 0) {
#pragma unroll
      for (int m = 0; m < MR; ++m) part[wave][m] = acc[m];
    }
    __syncthreads();
    if (ks != 0) return;
#pragma unroll
    for (int m = 0; m < MR; ++m)
#pragma unroll
      for (int s2 = 1; s2 < KS; ++s2) acc[m] += part[wave + s2][m];
  }
  if (lane != 0 || !col_ok) return;
  const float b = p.bias ? p.bias[col] : 0.f;
#pragma unroll
  for (int m = 0; m < MR; ++m) {
    if (m >= p.M) continue;
    float v = acc[m] + b;
    switch (p.act) {
      case ACT_SILU: v = v / (1.0f + expf(-v)); break;
      case ACT_RELU: v = fmaxf(v, 0.f); break;
      default: break;
    }
    v *= p.alpha;
    if (p.R) v += p.R[(size_t)m * p.ldr + col];
    p.C[(size_t)m * p.ldc + col] = v;
  }
}

bool gemv_eligible(const GemmArgs& a) {
  return smallm_eligible(a) && !a.glu && a.M <= 4 && a.Cin % 256 == 0 && a.Cin <= 2048;   // (the GEMV has no GLU epilogue)
}

template <int KS>
static int launch_gemv_ks(const GemmArgs& a, hipStream_t stream) {
  dim3 grid(cdiv(a.N, 4 / KS));
  ProfRec rec{}; bool prof = false;
  int rc = prof_begin(a, stream, 12, rec, prof);     // profiled with the small-M class (decode GEMVs)
  if (rc != SS_OK) return rc;
  if (a.ln_g) hipLaunchKernelGGL((gemv_kernel<4, KS, true>), grid, dim3(256), 0, stream, a);
  else hipLaunchKernelGGL((gemv_kernel<4, KS, false>), grid, dim3(256), 0, stream, a);
  SS_LAUNCH_CHECK();
  return prof_end(stream, rec, prof);
}

static int launch_gemv(const GemmArgs& a, hipStream_t stream) {
  // waves = N * KS: cover ~1024 SIMDs; KS must divide K/256
  const int kq = a.Cin / 256;
  if (a.N * 1 >= 2048 || kq % 2 != 0) return launch_gemv_ks<1>(a, stream);
  if (a.N * 2 >= 2048 || kq % 4 != 0) return launch_gemv_ks<2>(a, stream);
  return launch_gemv_ks<4>(a, stream);
}

// ---- optional event profiler -------------------------------------------------------------------
static const char* kTileNames[kNumTileCfg] = {
    "conv_gemm<128,16,16,4,1>", "conv_gemm<128,32,32,4,1>", "conv_gemm<128,32,16,4,1>", "conv_gemm<16,128,32,1,4>",
    "conv_gemm<16,128,16,1,4>", "conv_gemm<128,128,16,2,2>", "conv_gemm<128,64,32,2,2>", "conv_gemm<64,64,32,2,2>",
    "conv_gemm<64,64,16,2,2>", "conv_gemm<32,64,32,2,2>", "conv_gemm<32,64,16,2,2>", "conv_gemm<32,32,32,2,2>",
    "smallm_gemm<4,1>", "smallm_gemm<2,2>", "smallm_gemm<1,4>", "conv_sk<128,BN,32>",
    "conv_slab<32>", "conv_slab<16>", "conv_sk2<256,128,32>", "conv_sk2_bf16x3<256,128,32>",
    // whole-ResBlock launches of the narrow vocoder stages (resblock.hip) and the fused encoder FFN (ffn.hip): kernels of their own,
    // booked under their own names (round 3 booked resblock_fused under conv_slab<..>: VERDICT r3 "mislabelled second kernel")
    "resblock_fused<32>", "resblock_fused<16>", "ffn_fused<256,2048>", "rt_linear<48,256>", "conv_c64<256,64>", "conv_c32<256,32>", "conv_c16<256,16>", "conv_c64w<256,64>", "conv_c128w<256,128>", "conv_c32w<256,32>", "conv_c256w<256,128>", "rt_linear_kb<48,256>"};
static std::atomic<int> g_prof_mask{0};     // event brackets per class: written by ss_prof_enable between regions, read by every launch
static std::vector<ProfRec> g_prof_recs;
static std::vector<std::pair<hipEvent_t, hipEvent_t>> g_prof_pool;
static std::mutex g_prof_mu;

void prof_enable(int cls_mask) { g_prof_mask.store(cls_mask, std::memory_order_relaxed); }
const char* prof_cfg_name(int cls) { return (cls >= 0 && cls < kNumTileCfg) ? kTileNames[cls] : "?"; }
void prof_reset() {
  for (auto& r : g_prof_recs) g_prof_pool.push_back({r.e0, r.e1});
  g_prof_recs.clear();
}
int prof_read_issued(int cls, double* issued) {
  double v = 0;
  for (auto& r : g_prof_recs) if (r.cls == cls) v += r.issued;
  if (issued) *issued = v;
  return SS_OK;
}
int prof_read(int cls, double* ms_total, double* flops_total, long long* launches, double* bytes_total) {
  double ms = 0, fl = 0, by = 0; long long n = 0;
  for (auto& r : g_prof_recs) {
    if (r.cls != cls) continue;
    if (hipEventSynchronize(r.e1) != hipSuccess) return SS_ERR_HIP;
    float t = 0.f;
    if (hipEventElapsedTime(&t, r.e0, r.e1) != hipSuccess) return SS_ERR_HIP;
    ms += t; fl += r.flops; by += r.bytes; ++n;
  }
  if (ms_total) *ms_total = ms;
  if (flops_total) *flops_total = fl;
  if (launches) *launches = n;
  if (bytes_total) *bytes_total = by;
  return SS_OK;
}

// Always-on launch census (no events, no synchronisation): launches, algorithmic FLOPs and bytes per tile class since the
// library was loaded.  Lets a profile of a WHOLE process (rocprofv3 kernel stats, PMC passes) be divided by the
// algorithmic work of exactly the launches it saw.
// (relaxed atomics: eight host threads launch concurrently in bench.py; integer FLOP / byte counts -- 2^64 is ~18 EFLOP)
struct ProfTotals { std::atomic<unsigned long long> flops{0}, bytes{0}, launches{0}; };
static ProfTotals g_prof_totals[kNumTileCfg];
static std::mutex g_tot_mu;          // SS_SHAPE_LOG table only
// SS_SHAPE_LOG=<path>: per-(class, N, taps, Cin, operands) launch table written at process exit (tuning aid: which
// layers land on which kernel)
struct ShapeTot { long launches = 0; double rows = 0, flops = 0, bytes = 0; };
using ShapeKey = std::tuple<int, int, int, int, int>;
static std::map<ShapeKey, ShapeTot>* g_shapes = nullptr;
static std::string shape_table() {
  std::string out = "class N taps Cin operands launches mean_rows gflop_per_launch mbyte_per_launch\n";
  if (!g_shapes) return out;
  char line[256];
  for (auto& kv : *g_shapes) {
    const ShapeTot& z = kv.second;
    snprintf(line, sizeof line, "%d %d %d %d %d %ld %.0f %.3f %.2f\n", std::get<0>(kv.first), std::get<1>(kv.first), std::get<2>(kv.first),
             std::get<3>(kv.first), std::get<4>(kv.first), z.launches, z.rows / z.launches, z.flops / z.launches * 1e-9, z.bytes / z.launches * 1e-6);
    out += line;
  }
  return out;
}
static void shape_log_dump() {
  const char* path = getenv("SS_SHAPE_LOG");
  FILE* f = path ? fopen(path, "w") : nullptr;
  if (!f) return;
  std::lock_guard<std::mutex> lk(g_tot_mu);
  fputs(shape_table().c_str(), f);
  fclose(f);
}
static std::atomic<int> g_shape_log{getenv("SS_SHAPE_LOG") != nullptr ? 1 : 0};
void prof_shape_log(int on) {
  std::lock_guard<std::mutex> lk(g_tot_mu);
  if (on && g_shapes && !getenv("SS_SHAPE_LOG")) g_shapes->clear();     // a fresh in-process collection
  g_shape_log.store(on ? 1 : 0, std::memory_order_relaxed);
}
int prof_shape_dump(char* buf, int cap) {
  std::lock_guard<std::mutex> lk(g_tot_mu);
  const std::string t = shape_table();
  if (buf && cap > 0) { const int n = (int)std::min<size_t>(t.size(), (size_t)cap - 1); memcpy(buf, t.data(), n); buf[n] = 0; }
  return (int)t.size() + 1;
}
int prof_totals(int cls, double* flops, double* bytes, long long* launches) {
  if (cls < 0 || cls >= kNumTileCfg) return SS_ERR_ARG;
  if (flops) *flops = (double)g_prof_totals[cls].flops.load(std::memory_order_relaxed);
  if (bytes) *bytes = (double)g_prof_totals[cls].bytes.load(std::memory_order_relaxed);
  if (launches) *launches = (long long)g_prof_totals[cls].launches.load(std::memory_order_relaxed);
  return SS_OK;
}
static void algo_work(const GemmArgs& a, double& flops, double& bytes) {
  flops = a.algo_flops > 0 ? a.algo_flops : 2.0 * (double)a.M * a.N * a.taps * a.Cin;
  // algorithmic bytes: weights + bias once, every input row once, every output element once,
  // residual operands once (all f32); a second (pre-activated) output is NOT algorithmic
  const double ncols = a.glu ? a.N / 2 : a.N;
  bytes = 4.0 * ((double)a.N * a.taps * a.Cin + (a.bias ? a.N : 0) + (double)a.in_len * a.Cin +
                 (double)a.M * ncols * (1 + (a.R ? 1 : 0) + (a.R2 ? 1 : 0)));
  if (a.algo_bytes > 0) bytes = a.algo_bytes;
}

int prof_begin(const GemmArgs& a, hipStream_t stream, int cls, ProfRec& rec, bool& prof) {
  {
    double fl, by;
    algo_work(a, fl, by);
    g_prof_totals[cls].flops.fetch_add((unsigned long long)(fl + 0.5), std::memory_order_relaxed);
    g_prof_totals[cls].bytes.fetch_add((unsigned long long)(by + 0.5), std::memory_order_relaxed);
    g_prof_totals[cls].launches.fetch_add(1ull, std::memory_order_relaxed);
    if (g_shape_log.load(std::memory_order_relaxed)) {
      std::lock_guard<std::mutex> lk(g_tot_mu);
      if (!g_shapes) { g_shapes = new std::map<ShapeKey, ShapeTot>(); if (getenv("SS_SHAPE_LOG")) atexit(shape_log_dump); }
      const int ops = (a.R ? 1 : 0) | (a.R2 ? 2 : 0) | (a.C2 ? 4 : 0) | (a.in_act != ACT_NONE ? 8 : 0) | (a.act != ACT_NONE ? 16 : 0) | (a.glu ? 32 : 0) | (a.nseg > 0 ? 64 : 0);
      ShapeTot& z = (*g_shapes)[ShapeKey(cls, a.N, a.taps, a.Cin, ops)];
      z.launches += 1; z.rows += a.M; z.flops += fl; z.bytes += by;
    }
  }
  prof = ((unsigned)g_prof_mask.load(std::memory_order_relaxed) >> cls) & 1u;
  if (!prof) return SS_OK;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  if (!g_prof_pool.empty()) { rec.e0 = g_prof_pool.back().first; rec.e1 = g_prof_pool.back().second; g_prof_pool.pop_back(); }
  else { SS_HIP_CHECK(hipEventCreate(&rec.e0)); SS_HIP_CHECK(hipEventCreate(&rec.e1)); }
  rec.cls = cls;
  algo_work(a, rec.flops, rec.bytes);
  // Winograd classes (conv_c64w / conv_c128w / conv_c32w / conv_c256w): 4 MFMA k-blocks per output pair and full tap group + 2 (3) for a
  // last group of one (two) taps, instead of 2 k: 4 / 10 / 15 against 6 / 14 / 22 at k = 3 / 7 / 11
  rec.issued = (cls >= 27 && cls <= 30 && a.taps >= 3) ? rec.flops * (4.0 * (a.taps / 3) + (a.taps % 3 ? a.taps % 3 + 1 : 0)) / (2.0 * a.taps) : rec.flops;
  SS_HIP_CHECK(hipEventRecord(rec.e0, stream));
  return SS_OK;
}
int prof_end(hipStream_t stream, ProfRec& rec, bool prof) {
  if (prof) {
    // both event records of a bracket are made under the lock (it did not cure rocprofv3 7.2's crashes on the 8-thread bench
    // command -- those are inside its own launch interception, profiles/README.md -- but there is no reason to race here)
    std::lock_guard<std::mutex> lk(g_prof_mu);
    SS_HIP_CHECK(hipEventRecord(rec.e1, stream));
    g_prof_recs.push_back(rec);
  }
  return SS_OK;
}

// register prefetch depth: measured flat from 1 to 4 on MI355X (profiles/r01_tile_sweep.txt) -> 1
constexpr int default_pd(int, int) { return 1; }

template <int BM, int BN, int BK, int WM, int WN, int KS = 1, int PD = default_pd(BM, BN), bool BLK = false>
static int launch_cfg(const GemmArgs& a, hipStream_t stream, int cls) {
  constexpr size_t kLds = (size_t)KS * 2 * (BM + BN) * (BK + 8) * sizeof(float);
  static_assert(kLds <= 160 * 1024, "LDS budget");
  static_assert((size_t)(KS - 1) * BM * BN * sizeof(float) <= kLds, "reduction scratch fits the staging buffers");
  if constexpr (kLds > 64 * 1024) SS_MAX_LDS_ONCE((&conv_gemm_kernel<BM, BN, BK, WM, WN, KS, PD, BLK>), kLds);
  const int mmax = (a.nseg > 0 ? a.max_seg_out : a.M) - a.m_begin;
  dim3 grid(cdiv(mmax, BM), cdiv(a.N, BN), a.nseg > 0 ? a.nseg : 1);
  // XCD affinity (speed only): blocks land on XCD (linear id % 8).  With a multiple of 8 m-tiles per grid row every n-tile
  // column keeps m-tile i on XCD i % 8, so each XCD's L2 fetches 1/8 of the A rows (and all of W) instead of all of both --
  // the 3.0-3.9x counter-over-algorithmic traffic of the K = 256 encoder GEMMs (profiles/r02_pmc_traffic.json) was exactly
  // 8 x (A + W) + C.  The padding blocks leave at once (m0 >= out_len); not worth it for a handful of tiles.
  if (grid.x >= 32 && grid.y * grid.z > 1) grid.x = (grid.x + 7) & ~7u;
  ProfRec rec{}; bool prof = false;
  int rc = prof_begin(a, stream, cls, rec, prof);
  if (rc != SS_OK) return rc;
  hipLaunchKernelGGL((conv_gemm_kernel<BM, BN, BK, WM, WN, KS, PD, BLK>), grid, dim3(256 * KS), kLds, stream, a);
  SS_LAUNCH_CHECK();
  return prof_end(stream, rec, prof);
}

// pick the k-split so that tiles * 4 * KS waves cover the 1024 SIMDs (each group keeps >= 2 k-steps)
template <int BM, int BN, int BK, int WM, int WN, int MAXKS>
static int launch_cfg_ks(const GemmArgs& a, hipStream_t stream, int cls, long tiles) {
  const int nk = a.taps * (a.Cin / BK);
  int ks = 1;
  while (ks < MAXKS && tiles * 4 * ks < 1024 && nk / (ks * 2) >= 2) ks *= 2;
  if constexpr (MAXKS >= 4) { if (ks >= 4) return launch_cfg<BM, BN, BK, WM, WN, 4>(a, stream, cls); }
  if constexpr (MAXKS >= 2) { if (ks >= 2) return launch_cfg<BM, BN, BK, WM, WN, 2>(a, stream, cls); }
  return launch_cfg<BM, BN, BK, WM, WN, 1>(a, stream, cls);
}

template <int SK, int WN, bool GLU = false>
static int launch_smallm(const GemmArgs& a, hipStream_t stream, int cls) {
  dim3 grid((cdiv(a.N, 16 * WN) + 7) & ~7, cdiv(a.M, 16));   // see the kernel: n-tiles fastest, a multiple of 8 of them
  ProfRec rec{}; bool prof = false;
  int rc = prof_begin(a, stream, cls, rec, prof);
  if (rc != SS_OK) return rc;
  if (a.ln_g && a.Cin > 512) return SS_ERR_ARG;   // fused LayerNorm keeps the row in registers (D <= 512)
  if (a.ln_g) hipLaunchKernelGGL((smallm_gemm_kernel<SK, WN, true, GLU>), grid, dim3(256), 0, stream, a);
  else hipLaunchKernelGGL((smallm_gemm_kernel<SK, WN, false, GLU>), grid, dim3(256), 0, stream, a);
  SS_LAUNCH_CHECK();
  return prof_end(stream, rec, prof);
}

// (Dispatch::sk_min_flops, SS_SK_MIN_GFLOP: below this the small-tile kernels win -- tools/conv_bench.py sk)
// ---- stream-K workspaces (see gemm.hpp) -----------------------------------------------------------
constexpr size_t SKW_SYNC_BYTES = (16 + 1024) * sizeof(unsigned) + 256;      // >= both kernels' flag tables
static std::mutex g_skw_mu;
static std::vector<SkWorkspace*> g_skw_live;                                 // for the error count only
static std::map<std::pair<int, hipStream_t>, SkWorkspace*> g_skw_fallback;  // launches outside any context scope
static thread_local SkWorkspace* t_skw = nullptr;

SkWorkspace* sk_workspace_new() {
  SkWorkspace* w = new SkWorkspace();
  std::lock_guard<std::mutex> lk(g_skw_mu);
  g_skw_live.push_back(w);
  return w;
}
void sk_workspace_free(SkWorkspace* w) {
  if (!w) return;
  {
    std::lock_guard<std::mutex> lk(g_skw_mu);
    for (size_t i = 0; i < g_skw_live.size(); ++i)
      if (g_skw_live[i] == w) { g_skw_live.erase(g_skw_live.begin() + i); break; }
  }
  if (w->ws) (void)hipFree(w->ws);
  if (w->sync1) (void)hipFree(w->sync1);
  if (w->sync2) (void)hipFree(w->sync2);
  if (w->sync3) (void)hipFree(w->sync3);
  if (w->dbg) (void)hipFree(w->dbg);
  delete w;
}
// ---- dispatch settings (dispatch.hpp) ---------------------------------------------------------------
namespace {
std::mutex g_disp_mu;
std::atomic<unsigned> g_disp_gen{1};
Dispatch env_defaults() {
  Dispatch d;
  auto I = [](const char* k, int dflt) { const char* e = getenv(k); return e ? atoi(e) : dflt; };
  auto L = [](const char* k, long long dflt) { const char* e = getenv(k); return e ? atoll(e) : dflt; };
  d.attn_split = I("SS_ATTN_NO_SPLIT", 0) ? -1 : 0; d.attn_q16 = I("SS_ATTN_Q16", d.attn_q16);
  d.c16_off = I("SS_NO_CONV_C16", 0) ? 1 : 0; d.c32_off = I("SS_NO_CONV_C32", 0) ? 1 : 0; d.c64_off = I("SS_NO_CONV_C64", 0) ? 1 : 0;
  d.c16_min_rows = L("SS_CONV_C16_MIN_ROWS", d.c16_min_rows); d.c32_min_rows = L("SS_CONV_C32_MIN_ROWS", d.c32_min_rows);
  d.c64_min_rows = L("SS_CONV_C64_MIN_ROWS", d.c64_min_rows);
  d.c64w_on = I("SS_CONV_C64_WINOGRAD", 1); d.c128w_on = I("SS_CONV_C128_WINOGRAD", 1); d.c256w_on = I("SS_CONV_C256_WINOGRAD", 1);
  d.c32w_on = I("SS_CONV_C32_WINOGRAD", 1); d.c64w_min_k = I("SS_CONV_C64_WINOGRAD_MIN_K", 3);
  d.c128w_min_rows = L("SS_CONV_C128_MIN_ROWS", d.c128w_min_rows); d.c256w_min_rows = L("SS_CONV_C256_MIN_ROWS", d.c256w_min_rows);
  if (getenv("SS_FFN_WM")) { d.ffn_wm = I("SS_FFN_WM", 3); d.ffn_wm_forced = 1; }
  d.ffn_fusion = I("SS_NO_FFN_FUSION", 0) ? 0 : 1; d.ffn_min_rows = I("SS_FFN_MIN_ROWS", d.ffn_min_rows);
  if (getenv("SS_SK_MIN_GFLOP")) d.sk_min_flops = 1e9 * atof(getenv("SS_SK_MIN_GFLOP"));
  d.no_resblock_fusion = I("SS_NO_RESBLOCK_FUSION", 0); d.no_pair_fusion = I("SS_NO_PAIR_FUSION", 0);
  d.rt_off = I("SS_NO_RTLIN", 0) ? 1 : 0; d.rt_min_rows = I("SS_RTLIN_MIN_ROWS", d.rt_min_rows); d.rt_min_units = L("SS_RTLIN_MIN_UNITS", d.rt_min_units);
  d.rt_kb_min_units = L("SS_RTLIN_KB_MIN_UNITS", d.rt_kb_min_units); d.rt_kb_uw = I("SS_RTLIN_KB_UW", d.rt_kb_uw); d.rt_kb_xmap = I("SS_RTLIN_KB_XMAP", d.rt_kb_xmap);
  return d;
}
Dispatch& process_settings() { static Dispatch d = env_defaults(); return d; }      // (callers hold g_disp_mu)
thread_local const Dispatch* t_disp = nullptr;
thread_local CtxDispatch t_disp_own;          // launches outside any context (ss_op_* unit-test entry points)
}  // namespace
void dispatch_edit(const std::function<void(Dispatch&)>& fn) {
  std::lock_guard<std::mutex> lk(g_disp_mu);
  fn(process_settings());
  g_disp_gen.fetch_add(1, std::memory_order_release);
}
const Dispatch* CtxDispatch::refresh() {
  const unsigned now = g_disp_gen.load(std::memory_order_acquire);
  if (gen != now) {
    std::lock_guard<std::mutex> lk(g_disp_mu);
    d = process_settings();
    gen = g_disp_gen.load(std::memory_order_relaxed);
  }
  return &d;
}
const Dispatch& disp() { return t_disp ? *t_disp : *t_disp_own.refresh(); }
DispatchScope::DispatchScope(const Dispatch* d) : prev(t_disp) { t_disp = d; }
DispatchScope::~DispatchScope() { t_disp = prev; }

SkScope::SkScope(SkWorkspace* w) : prev(t_skw), prev_disp(t_disp) { t_skw = w; t_disp = w ? w->disp.refresh() : nullptr; }
SkScope::~SkScope() { t_skw = prev; t_disp = prev_disp; }

int sk_workspace_acquire(hipStream_t stream, SkWorkspace** out) {
  *out = nullptr;
  int dev = 0;
  SS_HIP_CHECK(hipGetDevice(&dev));
  SkWorkspace* w = t_skw;
  if (!w) {
    std::lock_guard<std::mutex> lk(g_skw_mu);
    SkWorkspace*& slot = g_skw_fallback[std::make_pair(dev, stream)];
    if (!slot) { slot = new SkWorkspace(); g_skw_live.push_back(slot); }
    w = slot;
  }
  if (!w->ws) {
    int cus = 0;
    SS_HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    if (cus <= 0) cus = 256;
    if (cus > 512) cus = 512;
    w->dev = dev; w->cus = cus;
    SS_HIP_CHECK(hipMalloc(&w->ws, (size_t)cus * 256 * 128 * sizeof(float)));
    SS_HIP_CHECK(hipMalloc(&w->sync1, SKW_SYNC_BYTES));
    SS_HIP_CHECK(hipMalloc(&w->sync2, SKW_SYNC_BYTES));
    SS_HIP_CHECK(hipMemsetAsync(w->sync1, 0, SKW_SYNC_BYTES, stream));
    SS_HIP_CHECK(hipMemsetAsync(w->sync2, 0, SKW_SYNC_BYTES, stream));
    SS_HIP_CHECK(hipMalloc(&w->sync3, 4096 * sizeof(unsigned)));
    SS_HIP_CHECK(hipMemsetAsync(w->sync3, 0, 4096 * sizeof(unsigned), stream));
    SS_HIP_CHECK(hipStreamSynchronize(stream));   // once per context: a later call may arrive on another stream and must see the zeros
  } else if (w->dev != dev) {
    return SS_ERR_ARG;                 // a context belongs to the device it first ran on
  }
  *out = w;
  return SS_OK;
}

int sk_workspace_error_count() {
  std::lock_guard<std::mutex> lk(g_skw_mu);
  int total = 0;
  for (SkWorkspace* w : g_skw_live)
    for (unsigned* sy : {w->sync1, w->sync2}) {
      unsigned v = 0;
      if (sy && hipMemcpy(&v, sy + 8, sizeof(v), hipMemcpyDeviceToHost) == hipSuccess) total += (int)v;
    }
  return total;
}

void debug_force_tile(int bm, int bn, int ks) {
  dispatch_edit([=](Dispatch& d) { d.force_bm = bm; d.force_bn = bn; d.force_ks = ks; d.sk_groups = (bm == 1 && bn == 8); });
}
bool debug_tile_forced() { return disp().force_bm != 0; }

static thread_local int t_canon = CANON_NONE;
CanonScope::CanonScope(int mode) : prev(t_canon) { t_canon = mode; }
CanonScope::~CanonScope() { t_canon = prev; }
int canon_mode() { return t_canon; }
void canon_debug_set(int mode) { t_canon = mode; }

// CANON_SMALLM: what the no-LDS kernel computes at all (smallm_eligible minus the tuning hooks)
static bool smallm_shape_ok(const GemmArgs& a) {
  const bool glu_ok = !a.glu || (a.N % 32 == 0 && a.act == ACT_NONE && a.alpha == 1.f && !a.R);
  return a.taps == 1 && a.stride == 1 && a.pad == 0 && glu_ok && a.nseg == 0 && a.chunk == 0 && a.in_act == ACT_NONE && !a.R2 &&
         !a.C2 && a.div == 0.f && !a.ln_out && (a.act == ACT_NONE || a.act == ACT_SILU || a.act == ACT_RELU) && a.M > 0 &&
         a.M <= 1024 && a.Cin % 64 == 0 && (a.lda & 3) == 0 && (!a.ln_g || a.Cin <= 512);     // (rows = utterances of a lock-step pack: 16-row tiles are independent, any count gives a row the same bits)
}

// Pack-invariant routes (GemmArgs::canon): see gemm.hpp.  Nothing here may make the BITS depend on M; the choice among kernels
// that give the same bits may.
static int launch_canon(const GemmArgs& a, hipStream_t stream) {
  const int M = a.nseg > 0 ? a.max_seg_out : a.M;
  const bool k32 = (a.Cin % 32) == 0;
  const int nseg = a.nseg > 0 ? a.nseg : 1;
  if (a.ln_out) return SS_ERR_ARG;
  if (a.canon == CANON_SMALLM) {
    if (!smallm_shape_ok(a)) return SS_ERR_ARG;
    if (a.glu) return launch_smallm<2, 2, true>(a, stream, 13);
    // split-K form by (N, K) alone: 4 waves on one 16-column tile's k-quarters, or 2 x 2 when N alone fills the chip
    if (a.Cin >= 1024 || a.N <= 4096) return launch_smallm<4, 1>(a, stream, 12);
    return launch_smallm<2, 2>(a, stream, 13);
  }
  // ---- CANON_SEQ: one accumulator chain per output element ----
  if (rtlin_shape_ok(a) && (a.ln_g || rtlin_eligible(a))) return launch_rtlin(a, stream);
  if (a.ln_g) return SS_ERR_ARG;            // LayerNorm prologue: the row-tile kernel only (K = 256); callers normalise first otherwise
  if (rtlin_kb_eligible(a)) return launch_rtlin_kb(a, stream);      // K = 512 ... : row tile through LDS per 256-wide k-block (same bits as the BLK tiles below)
  // (rounds 4-5 sent big N % 128 == 0 shapes to conv_sk2 cut on whole tiles -- ONE chain over all of K; its 128 accumulator
  //  registers per wave leave no room for the second accumulator set the blocked chain needs, so CANON_SEQ no longer routes there)
  // LDS-tiled kernel, one k-group, blocked chain (BLK)
  if (a.N <= 16) return launch_cfg<128, 16, 16, 4, 1, 1, 1, true>(a, stream, 0);
  if (a.N <= 32 && !a.glu) return k32 ? launch_cfg<128, 32, 32, 4, 1, 1, 1, true>(a, stream, 1) : launch_cfg<128, 32, 16, 4, 1, 1, 1, true>(a, stream, 2);
  if (M <= 16) return k32 ? launch_cfg<16, 128, 32, 1, 4, 1, 1, true>(a, stream, 3) : launch_cfg<16, 128, 16, 1, 4, 1, 1, true>(a, stream, 4);
  const long t3264 = (long)cdiv(M, 32) * cdiv(a.N, 64) * nseg;
  if (!k32) return launch_cfg<32, 64, 16, 2, 2, 1, 1, true>(a, stream, 10);
  if (a.glu || t3264 >= 768) return launch_cfg<32, 64, 32, 2, 2, 1, 1, true>(a, stream, 9);
  return launch_cfg<32, 32, 32, 2, 2, 1, 1, true>(a, stream, 11);
}

bool smallm_eligible(const GemmArgs& a) {
  if (disp().force_bm > 1) return false;
  const int M = a.nseg > 0 ? a.max_seg_out : a.M;
  // GLU (the conformer conv module's pointwise conv 1): only in its plain form -- no activation, scale or residual on top
  const bool glu_ok = !a.glu || (a.N % 32 == 0 && a.act == ACT_NONE && a.alpha == 1.f && !a.R);
  const bool plain_linear = a.taps == 1 && a.stride == 1 && a.pad == 0 && glu_ok && a.nseg == 0 && a.chunk == 0 &&
                            a.in_act == ACT_NONE && !a.R2 && !a.C2 && a.div == 0.f &&
                            (a.act == ACT_NONE || a.act == ACT_SILU || a.act == ACT_RELU);
  // row limit of the no-LDS kernel: every 16-row tile re-streams its W columns from L2, so it only pays while the whole
  // problem is latency-bound (SS_SMALLM_MAX_ROWS: tuning knob, tools/latency_breakdown.py)
  // 192: a single utterance of 5-7.7 s (T' = 129..192 encoder rows, a third of the CVSS-C length distribution) keeps its 12 x 9
  // linears on this kernel instead of the 32x32-tile kernel, whose k-loop is one global-load round trip per k-step when a
  // launch is this small: 6.43 -> 6.26 ms per utterance at T' = 131 (tools/b1_profile.py); beyond ~200 rows the re-streaming
  // costs what the tiles' latency did (8.24 vs 8.18 ms at T' = 201 with 256), and the 500-row unit decoder loses 2x.
  static const int max_rows = getenv("SS_SMALLM_MAX_ROWS") ? atoi(getenv("SS_SMALLM_MAX_ROWS")) : 192;
  return plain_linear && M <= max_rows && M > 0 && a.Cin % 64 == 0 && (a.lda & 3) == 0;
}


int launch_conv_gemm(const GemmArgs& a_in, hipStream_t stream) {
  GemmArgs a = a_in;
  const int M = a.nseg > 0 ? a.max_seg_out : a.M;
  if (M <= 0 || a.N <= 0) return SS_OK;
  if (a.Cin % 16 != 0 || (a.lda & 3) != 0 || a.taps < 1) return SS_ERR_ARG;
  if (a.glu && (a.N % 32 != 0 || a.C2)) return SS_ERR_ARG;
  const bool k32 = (a.Cin % 32) == 0;
  const int nseg = a.nseg > 0 ? a.nseg : 1;
  if (a.canon == CANON_NONE) a.canon = t_canon;
  if (a.canon != CANON_NONE && !disp().force_bm) { a.m_begin = 0; return launch_canon(a, stream); }
  if (a.m_begin > 0) {       // a row range of a strided conv (ss_encoder_stream_forward): only the LDS-tile kernel starts anywhere but row 0
    if (a.nseg > 0 || a.m_begin >= a.M || a.ln_g || a.ln_out || a.N <= 32) return SS_ERR_ARG;
    const long t = (long)cdiv(a.M - a.m_begin, 32) * cdiv(a.N, 64);
    if (k32 && t >= 768) return launch_cfg<32, 64, 32, 2, 2, 1>(a, stream, 9);
    return k32 ? launch_cfg_ks<32, 64, 32, 2, 2, 4>(a, stream, 9, t) : launch_cfg_ks<32, 64, 16, 2, 2, 4>(a, stream, 10, t);
  }
  // a forced tile (tuning hook) keeps M <= 4 launches off the GEMV -- except when the caller asked for ln_out, which only the GEMV writes
  if ((!disp().force_bm || a.ln_out) && gemv_eligible(a)) return launch_gemv(a, stream);
  if (a.ln_out) return SS_ERR_ARG;      // only the GEMV form publishes the normalised rows
  if (smallm_eligible(a)) {
    if (a.glu) return launch_smallm<2, 2, true>(a, stream, 13);   // value / gate tiles side by side in one workgroup
    const long wgs16 = (long)cdiv(M, 16) * cdiv(a.N, 16);
    if (a.Cin >= 1024 || wgs16 <= 1024) return launch_smallm<4, 1>(a, stream, 12);
    if (wgs16 <= 4096) return launch_smallm<2, 2>(a, stream, 13);
    return launch_smallm<1, 4>(a, stream, 14);
  }
  // 64-channel vocoder stage of a packed batch: input slab in LDS once, W fragments from L2 (conv_c64.hip)
  if (!disp().force_bm && conv_c64_eligible(a)) return conv_c64w_eligible(a) ? launch_conv_c64w(a, stream) : launch_conv_c64(a, stream);
  if (!disp().force_bm && conv_c128w_eligible(a)) return launch_conv_c128w(a, stream);
  if (!disp().force_bm && conv_c256w_eligible(a)) return launch_conv_c256w(a, stream);
  if (!disp().force_bm && conv_c32_eligible(a)) return conv_c32w_eligible(a) ? launch_conv_c32w(a, stream) : launch_conv_c32(a, stream);
  if (!disp().force_bm && conv_c16_eligible(a)) return launch_conv_c16(a, stream);
  // K = 256 linears of packed batches (encoder projections, CTC heads, cross K|V): row tile in LDS, W fragments from L2 (rtlin.hip)
  if (!disp().force_bm && rtlin_eligible(a)) return launch_rtlin(a, stream);
  if (a.ln_g) return SS_ERR_ARG;  // LayerNorm fusion exists only on the small-M and row-tile paths
  if (disp().force_bm == 1 && conv_sk_eligible(a)) return launch_conv_sk(a, stream, disp().force_ks);   // tuning hook: stream-K, grid = ks (0 = auto)
  if (disp().force_bm == 4 && conv_sk2_eligible(a)) return launch_conv_sk2(a, stream, disp().force_ks);  // tuning hook: 2nd-generation stream-K
  if (disp().force_bm == 5 && conv_sk2_eligible(a)) { GemmArgs b = a; b.x3 = 1; return launch_conv_sk2(b, stream, disp().force_ks); }  // ... its split-bf16 variant (tests, tools/sk2_bench.py)
  // Big "same" convs / linears (packed vocoder batches, unit-decoder FFN): persistent stream-K
  // 128-wide tiles, 95-110 TFLOP/s against 75-90 for the 32x64 kernel (profiles/r01_sk_sweep.txt).
  // They need enough k-steps per workgroup to amortise the fix-up + epilogue: >= 12 at BN = 128;
  // at BN = 64 (half the MFMA work per k-step) only the k >= 7 convs qualify.
  if (!disp().force_bm && conv_sk_eligible(a) && 2.0 * (double)a.M * a.N * a.taps * a.Cin >= disp().sk_min_flops) {
    const long long nk = (long long)a.taps * (a.Cin / 32);
    const bool wide = a.N % 128 == 0;
    const long long units = (long long)cdiv(a.M, 128) * (a.N / (wide ? 128 : 64)) * nk;
    static const long long min_units = getenv("SS_SK_MIN_UNITS") ? atoll(getenv("SS_SK_MIN_UNITS")) : 12 * 512;   // tuning knob
    static const bool no_sk2 = getenv("SS_NO_SK2") && atoi(getenv("SS_NO_SK2"));   // A/B knob: first-generation kernel only
    static const int sk2_min_k64 = getenv("SS_SK2_MINK64") ? atoi(getenv("SS_SK2_MINK64")) : 192;   // N % 128 == 64: smallest taps * Cin for conv_sk2<64>
    if (!no_sk2 && conv_sk2_eligible(a) && (wide ? units >= min_units : a.taps * a.Cin >= sk2_min_k64)) return launch_conv_sk2(a, stream);
    if (wide ? units >= min_units : a.taps * a.Cin >= 448) return launch_conv_sk(a, stream);
  }
  if (disp().force_bm && a.N > 32 && k32) {   // tuning hook (tools/conv_bench.py): ks = KS*10 + PD
    const int f = disp().force_bm * 10000 + (disp().force_bn % 100) * 100 + disp().force_ks;   // 128x128 -> bn code 28... see cases
    switch (f) {
#define SS_CASE(BM_, BN_, KS_, PD_, CLS_) case BM_ * 10000 + BN_ * 100 + KS_ * 10 + PD_: return launch_cfg<BM_, BN_, 32, 2, 2, KS_, PD_>(a, stream, CLS_);
      SS_CASE(64, 64, 1, 1, 7) SS_CASE(64, 64, 1, 2, 7) SS_CASE(64, 64, 1, 3, 7)
      SS_CASE(64, 64, 2, 2, 7) SS_CASE(64, 64, 2, 3, 7)
      SS_CASE(32, 64, 1, 1, 9) SS_CASE(32, 64, 1, 2, 9) SS_CASE(32, 64, 1, 3, 9) SS_CASE(32, 64, 1, 4, 9)
      SS_CASE(32, 64, 2, 3, 9) SS_CASE(32, 64, 4, 3, 9)
      SS_CASE(32, 32, 1, 1, 11) SS_CASE(32, 32, 1, 2, 11) SS_CASE(32, 32, 1, 3, 11) SS_CASE(32, 32, 1, 4, 11)
      SS_CASE(32, 32, 2, 3, 11) SS_CASE(32, 32, 4, 3, 11)
#undef SS_CASE
      case 128 * 10000 + 28 * 100 + 11: return launch_cfg<128, 128, 16, 2, 2, 1, 1>(a, stream, 5);
      case 128 * 10000 + 64 * 100 + 11: return launch_cfg<128, 64, 32, 2, 2, 1, 1>(a, stream, 6);
      default: break;
    }
  }
  // narrow vocoder stages: weights + input slab in LDS, one HBM pass (conv_slab.hip)
  if (disp().force_bm != 2 && conv_slab_eligible(a) && a.M >= 2048) return launch_conv_slab(a, stream);
  if (a.N <= 16) return launch_cfg_ks<128, 16, 16, 4, 1, 2>(a, stream, 0, (long)cdiv(M, 128) * nseg);
  if (a.N <= 32 && !a.glu) {
    const long tl = (long)cdiv(M, 128) * nseg;
    return k32 ? launch_cfg_ks<128, 32, 32, 4, 1, 2>(a, stream, 1, tl) : launch_cfg_ks<128, 32, 16, 4, 1, 2>(a, stream, 2, tl);
  }
  if (M <= 16) {
    return k32 ? launch_cfg<16, 128, 32, 1, 4>(a, stream, 3) : launch_cfg<16, 128, 16, 1, 4>(a, stream, 4);
  }
  // Tile choice (tools/conv_bench.py sweep on MI355X, profiles/r01_tile_sweep.txt): these problems are
  // a few GFLOP at most, so what matters is an even spread over 1024 SIMDs with several resident
  // workgroups per CU -- small 32x32 tiles, plus intra-workgroup split-K when even those are few.
  const long t128 = (long)cdiv(M, 128) * cdiv(a.N, 128) * nseg;
  const long t3232 = (long)cdiv(M, 32) * cdiv(a.N, 32) * nseg;
  const long t3264 = (long)cdiv(M, 32) * cdiv(a.N, 64) * nseg;
  (void)t128;   // 128x128 tiles lose to 32x64 even at batch scale (profiles/r01_tile_sweep_batch.txt)
  if (k32 && t3264 >= 768) return launch_cfg<32, 64, 32, 2, 2, 1>(a, stream, 9);   // big (batched) problems: ~90 TFLOP/s
  if (!k32 || a.glu) {
    return k32 ? launch_cfg_ks<32, 64, 32, 2, 2, 4>(a, stream, 9, t3264) : launch_cfg_ks<32, 64, 16, 2, 2, 4>(a, stream, 10, t3264);
  }
  const int nk = a.taps * (a.Cin / 32);
  if (t3232 >= 700 || nk < 4) return launch_cfg<32, 32, 32, 2, 2, 1>(a, stream, 11);
  if (t3232 >= 300 || nk < 8) return launch_cfg<32, 32, 32, 2, 2, 2>(a, stream, 11);
  return launch_cfg<32, 32, 32, 2, 2, 4>(a, stream, 11);
}

}  // namespace ss
