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

namespace ss {

using f32x4 = __attribute__((ext_vector_type(4))) float;

template <int BM, int BN, int BK, int WM, int WN>
__global__ __launch_bounds__(256) void conv_gemm_kernel(const GemmArgs p) {
  static_assert(WM * WN == 4, "4 waves per workgroup");
  constexpr int LDK = BK + 4;  // +4 floats keeps 16-B alignment and staggers banks
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int TM = WTM / 16, TN = WTN / 16;
  static_assert(TM >= 1 && TN >= 1, "wave tile must hold a 16x16 MFMA tile");
  constexpr int KQ = BK / 4;
  constexpr int A_F4 = BM * KQ, W_F4 = BN * KQ;
  constexpr int NA = (A_F4 + 255) / 256, NW = (W_F4 + 255) / 256;
  constexpr int STAGE = (BM + BN) * LDK;
  __shared__ __attribute__((aligned(16))) float smem[2 * STAGE];

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int r = lane & 15, g = lane >> 4;

  int out_start = 0, out_len = p.M, in_start = 0, in_len = p.in_len;
  if (p.nseg > 0) {
    const int* s = p.segs + 4 * blockIdx.z;
    out_start = s[0]; out_len = s[1]; in_start = s[2]; in_len = s[3];
  }
  const int m0 = blockIdx.x * BM;
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

  f32x4 ra[NA], rw[NW];
  auto load_regs = [&](int kb) {
    const int tap = kb / kpt;
    const int ci0 = (kb - tap * kpt) * BK;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int rin = a_rin0[i] + tap * p.dil;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (a_ok[i] && rin >= 0 && rin < in_len && rin < a_lim[i]) {
        v = *reinterpret_cast<const f32x4*>(p.A + (size_t)(in_start + rin) * p.lda + ci0 + a_c4[i]);
        if (p.in_act == ACT_LRELU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * p.in_slope;
        }
      }
      ra[i] = v;
    }
#pragma unroll
    for (int i = 0; i < NW; ++i) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (w_ok[i]) v = *reinterpret_cast<const f32x4*>(p.W + w_off[i] + (size_t)kb * BK);
      rw[i] = v;
    }
  };
  auto store_lds = [&](int buf) {
    float* base = smem + buf * STAGE;
#pragma unroll
    for (int i = 0; i < NA; ++i)
      if (t + i * 256 < A_F4) *reinterpret_cast<f32x4*>(base + a_lds[i]) = ra[i];
#pragma unroll
    for (int i = 0; i < NW; ++i)
      if (t + i * 256 < W_F4) *reinterpret_cast<f32x4*>(base + w_lds[i]) = rw[i];
  };

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = {0.f, 0.f, 0.f, 0.f};

  load_regs(0);
  store_lds(0);
  __syncthreads();

  for (int kb = 0; kb < nk; ++kb) {
    const int cur = kb & 1;
    if (kb + 1 < nk) load_regs(kb + 1);
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
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i][e], bf[j][e], acc[i][j], 0, 0, 0);
    }
    if (kb + 1 < nk) store_lds(cur ^ 1);
    __syncthreads();
  }

  // ---- epilogue ----
  auto activate = [&](float v) -> float {
    switch (p.act) {
      case ACT_SILU: return v / (1.0f + expf(-v));
      case ACT_RELU: return fmaxf(v, 0.f);
      case ACT_TANH: return tanhf(v);
      default: return v;
    }
  };
  if (p.glu) {
    if constexpr (TN % 2 == 0) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; j += 2) {
          const int ncol = n0 + wn * WTN + j * 16;      // start of the [value|gate] 32-col block
          const int oc = ncol / 2 + r;                    // output channel
          if (ncol + 16 + r < p.N) {
            const float bv = p.bias ? p.bias[ncol + r] : 0.f;
            const float bg = p.bias ? p.bias[ncol + 16 + r] : 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int m = m0 + wm * WTM + i * 16 + g * 4 + e;
              if (m < out_len) {
                const float val = acc[i][j][e] + bv;
                const float gate = acc[i][j + 1][e] + bg;
                p.C[(size_t)(out_start + m) * p.ldc + oc] = val * (1.0f / (1.0f + expf(-gate)));
              }
            }
          }
        }
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + wn * WTN + j * 16 + r;
      if (n < p.N) {
        const float b = p.bias ? p.bias[n] : 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int m = m0 + wm * WTM + i * 16 + g * 4 + e;
          if (m < out_len) {
            const size_t row = (size_t)(out_start + m);
            float v = activate(acc[i][j][e] + b) * p.alpha;
            if (p.R) v += p.R[row * p.ldr + n];
            if (p.R2) v = p.R2[row * p.ldr2 + n] + v;
            if (p.div > 0.f) v = v / p.div;
            p.C[row * p.ldc + n] = v;
          }
        }
      }
    }
}

template <int BM, int BN, int BK, int WM, int WN>
static int launch_cfg(const GemmArgs& a, hipStream_t stream) {
  const int mmax = a.nseg > 0 ? a.max_seg_out : a.M;
  dim3 grid(cdiv(mmax, BM), cdiv(a.N, BN), a.nseg > 0 ? a.nseg : 1);
  hipLaunchKernelGGL((conv_gemm_kernel<BM, BN, BK, WM, WN>), grid, dim3(256), 0, stream, a);
  SS_LAUNCH_CHECK();
  return SS_OK;
}

int launch_conv_gemm(const GemmArgs& a, hipStream_t stream) {
  const int M = a.nseg > 0 ? a.max_seg_out : a.M;
  if (M <= 0 || a.N <= 0) return SS_OK;
  if (a.Cin % 16 != 0 || (a.lda & 3) != 0 || a.taps < 1) return SS_ERR_ARG;
  if (a.glu && (a.N % 32 != 0)) return SS_ERR_ARG;
  const bool k32 = (a.Cin % 32) == 0;
  const int nseg = a.nseg > 0 ? a.nseg : 1;
  if (a.N <= 16) return launch_cfg<128, 16, 16, 4, 1>(a, stream);
  if (a.N <= 32 && !a.glu) {
    return k32 ? launch_cfg<128, 32, 32, 4, 1>(a, stream) : launch_cfg<128, 32, 16, 4, 1>(a, stream);
  }
  if (M <= 16) {
    return k32 ? launch_cfg<16, 128, 32, 1, 4>(a, stream) : launch_cfg<16, 128, 16, 1, 4>(a, stream);
  }
  const long t128 = (long)cdiv(M, 128) * cdiv(a.N, 128) * nseg;
  const long t12864 = (long)cdiv(M, 128) * cdiv(a.N, 64) * nseg;
  const long t64 = (long)cdiv(M, 64) * cdiv(a.N, 64) * nseg;
  if (t128 >= 192) return launch_cfg<128, 128, 16, 2, 2>(a, stream);
  if (t12864 >= 192 && k32) return launch_cfg<128, 64, 32, 2, 2>(a, stream);
  if (t64 >= 160 || M > 256) {
    return k32 ? launch_cfg<64, 64, 32, 2, 2>(a, stream) : launch_cfg<64, 64, 16, 2, 2>(a, stream);
  }
  return k32 ? launch_cfg<32, 64, 32, 2, 2>(a, stream) : launch_cfg<32, 64, 16, 2, 2>(a, stream);
}

}  // namespace ss
