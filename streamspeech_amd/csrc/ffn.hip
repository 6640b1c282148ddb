// Fused Conformer feed-forward module for packed batches (gfx950, exact-f32 MFMA):
//   y = x + alpha * (W2 . SiLU(W1 . LayerNorm(x) + b1) + b2)      [optionally followed by the layer's final LayerNorm]
// in ONE persistent launch -- reference FeedForwardModule.forward + the macaron wiring of ChunkConformerEncoderLayer.forward
// (researches/chunk_unity/modules/conformer_layer.py:152-164, 254-312), SURVEY.md §8a row a4.
//
// Why a kernel of its own: as two GEMM launches the 256 -> 2048 half is a K = 256 problem -- 8 k-steps per
// tile, bound by tile prologue / epilogue and L2 -> LDS traffic at 0.36-0.46 of the FP32-MFMA peak -- and the [rows, 2048]
// hidden tensor (34 MB per FFN at 4200 packed rows) makes a round trip through HBM.  Here the hidden activations never leave
// the registers:
//   * work unit = (64-row tile, 16 hidden columns).  GEMM1 of a unit -- H[64 x 16] = LN(x)[64 x 256] . W1[16 x 256]^T -- is
//     issued with swapped operands (D = W . A^T), so its accumulator tile holds, per lane, 4 CONSECUTIVE hidden columns of ONE
//     row: exactly the operand fragment GEMM2 -- Y[64 x 256] += SiLU(H + b1)[64 x 16] . W2[256 x 16]^T -- needs (with the
//     k-permutation "k = 4 g + e" shared by both operands), so H goes from GEMM1's accumulators through bias + SiLU straight
//     into GEMM2's MFMAs: no LDS, no HBM.  512 MFMAs per unit per wave.
//   * the (tile, unit) space of a launch is cut into equal contiguous ranges, one per workgroup (one per CU) -- stream-K over
//     the hidden dimension, so the chip is full whatever the row count; inside a workgroup the four waves (one per SIMD) take
//     disjoint unit sub-ranges of the SAME 64 rows: the LayerNorm-ed row tile is computed once per part into LDS (64 KB, XOR-
//     swizzled 16-B chunks -> conflict-free ds_read_b128 fragments) and read by all four; weight fragments go from L2 straight
//     to registers (one 16-B row segment per lane, a ring of 8 requested 8 steps = 4096+ MFMA cycles ahead): a W fragment
//     feeds 16 MFMAs, an LDS fragment 4, there is no barrier and no LDS-DMA piece inside the contraction.
//   * the four waves' partial Y tiles are summed through LDS as a 3-round reduce-scatter (fixed order), after which wave w
//     owns columns [64 w, 64 w + 64) of the workgroup's partial.
//   * a row tile whose units are split over several workgroups: every contributor parks its [64 x 256] partial as sc1
//     (write-through) b128 stores, drains them, and bumps the tile's arrival counter; the LAST arrival -- nobody waits for
//     anybody, so nothing depends on residency or dispatch order -- adds the parked partials in workgroup order (its own from
//     registers at its place: fixed association, bit-reproducible), applies bias, alpha, the residual and (optionally) the
//     LayerNorm that follows the layer's second FFN, and resets the counter (guide §6 Guideline 16, counter form R1).
// Everything is exact f32 (v_mfma_f32_16x16x4_f32 = an fmaf chain); per row the result differs from the two-launch path only by
// the association order of the 2048-term hidden sum.
#include "gemm.hpp"

#include <cstdio>
#include <cstdlib>

// Without a fence per step hipcc sinks every weight load to just before its first use (one step = 400-500 cycles ahead instead
// of FF_RING steps): the loop then waits out an L2 round trip per step.  FF_FENCE=0 compiles the fences out (A/B).
#ifndef FF_FENCE
#define FF_FENCE 1
#endif
#if FF_FENCE
#define FF_STEP_FENCE __builtin_amdgcn_sched_barrier(0)
#else
#define FF_STEP_FENCE do { } while (0)
#endif

namespace ss {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

namespace {

constexpr int FF_D = 256;                     // model width: K of W1, N of W2
constexpr int FF_BM_MAX = 64;                 // rows per tile = 16 WMT (WMT = MFMA row tiles per wave: 4 | 3, template argument)
constexpr int FF_UN = 16;                     // hidden columns per unit
[[maybe_unused]] constexpr int FF_RING = 8;                    // weight fragments in flight per wave
[[maybe_unused]] constexpr int FF_SC1 = 16;                    // buffer cache policy bit: agent scope (write-through stores, L1-bypassing loads)
[[maybe_unused]] constexpr int FF_SLOT = FF_BM_MAX * FF_D;     // floats of one parked-partial slot (sized for 64-row tiles)
constexpr int FF_MAX_TILES = 4096;            // arrival counters per context
constexpr int FF_XS = FF_D + 4;                // LDS row stride of the LayerNorm-ed tile (floats): 260 = 4 mod 64 banks -> the 16-B chunk of lane
                                              // (r, g) sits in bank quad (r + g) & 15: one 2-way conflict per ds_read_b128, and every fragment
                                              // address is ONE per-lane base + an immediate (an XOR swizzle needs a VGPR per fragment: 64 of them)
[[maybe_unused]] constexpr int FF_NUM_RECORDS = 0x7ffffff0;
constexpr size_t ff_lds_bytes(int bm) { return (size_t)(bm * FF_XS + 2 * bm * 4 + 16) * sizeof(float); }

struct FfnKArgs {
  const float* X; int ldx;
  float* Y; int ldy;
  const float *ln_g, *ln_b, *W1, *b1, *W2, *b2, *ln2_g, *ln2_b;
  float alpha;
  int M, F, G;
  int canon;              // != 0: ranges end on tile boundaries -- every row tile is computed whole by one workgroup (pack-invariant bits)
  int zero;               // 0 at run time, opaque at compile time (see xoff in the unit loop)
  float* ws;              // [2 G][64 x 256] parked partials: slot 2 w (+1: the workgroup's second incomplete tile)
  unsigned* cnt;          // [tiles] arrival counters, zero between launches
};

}  // namespace

template <int WMT>
__global__ __launch_bounds__(256, WMT <= 2 ? 2 : 1) void ffn_fused_kernel(const FfnKArgs p) {
#if __HIP_DEVICE_COMPILE__
  constexpr int FF_BM = 16 * WMT;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* xs = smem;                                  // [64][260] LayerNorm-ed rows (padded stride, see FF_XS)
  float* red = smem + FF_BM * FF_XS;                 // [2][64 rows][4 waves] row statistics of the optional output LayerNorm
  int* s_misc = reinterpret_cast<int*>(red + 2 * FF_BM * 4);

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int r = lane & 15, g = lane >> 4;
  const int w = blockIdx.x;
  const int F = p.F, UT = F / FF_UN;
  const int tiles = (p.M + FF_BM - 1) / FF_BM;
  const long long U = (long long)tiles * UT;
  long long u0 = (long long)w * U / p.G, u1 = (long long)(w + 1) * U / p.G;
  if (p.canon) {        // whole tiles: wave w of the owner contracts hidden units [UT/4 w, UT/4 (w+1)) of every row, whatever M and G are
    u0 = ((long long)w * tiles / p.G) * UT; u1 = ((long long)(w + 1) * tiles / p.G) * UT;
  }
  if (u1 <= u0) return;
  const int t_first = (int)(u0 / UT), t_last = (int)((u1 - 1) / UT);

  for (int tile = t_first; tile <= t_last; ++tile) {
    const long long ut0 = (long long)tile * UT;
    const int ka = (int)(max(u0, ut0) - ut0), kb = (int)(min(u1, ut0 + UT) - ut0);
    const int m0 = tile * FF_BM;

    // ---- LayerNorm of the row tile into LDS (thread = row t / 4, chunks q, q + 4, ...: 64-B row segments per 4 lanes) ----
    __syncthreads();                                  // every wave is done with the previous part's use of xs / red
    if (t < 4 * FF_BM) {
      const int row = t >> 2, q = t & 3;
      const int m = min(m0 + row, p.M - 1);         // clamped (branch-free loads): rows >= M are computed on a copy and never stored
      f32x4 v[16];
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) v[i] = *reinterpret_cast<const f32x4*>(p.X + (size_t)m * p.ldx + (i * 4 + q) * 4);
#pragma unroll
      for (int i = 0; i < 16; ++i) s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
      s += __shfl_xor(s, 1, 64);
      s += __shfl_xor(s, 2, 64);
      const float mean = s * (1.0f / FF_D);
      float qq = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d = v[i][e] - mean; qq += d * d; }
      qq += __shfl_xor(qq, 1, 64);
      qq += __shfl_xor(qq, 2, 64);
      const float rstd = 1.0f / sqrtf(qq * (1.0f / FF_D) + 1e-5f);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int c = i * 4 + q;
        const f32x4 gm = *reinterpret_cast<const f32x4*>(p.ln_g + c * 4);
        const f32x4 bt = *reinterpret_cast<const f32x4*>(p.ln_b + c * 4);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (v[i][e] - mean) * rstd * gm[e] + bt[e];
        *reinterpret_cast<f32x4*>(xs + row * FF_XS + (c << 2)) = o;
      }
    }
    __syncthreads();

    // ---- this wave's units of the part: a contiguous quarter of [ka, kb) ----
    const int n_part = kb - ka;
    const int ubase = n_part >> 2, urem = n_part & 3;
    const int my_n = ubase + (wave < urem ? 1 : 0);
    const int my_u0 = ka + wave * ubase + (wave < urem ? wave : urem);

    f32x4 yacc[WMT][16];
#pragma unroll
    for (int i = 0; i < WMT; ++i)
#pragma unroll
      for (int c = 0; c < 16; ++c) yacc[i][c] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (my_n > 0) {
      // weight fragment of flat step s of the unit at hidden column n0: s < 16: W1 rows n0 .. n0+15, k-group s;
      // s >= 16: W2 rows 16 (s-16) .. +15, hidden columns n0 .. n0+15.  Lane (r, g) takes 16 B of row r at k = 4 g: buffer loads
      // with ONE per-lane offset per matrix and the (unit, step) part in the wave-uniform soffset (no address VGPRs per step).
      const __amdgpu_buffer_rsrc_t rsW1 = __builtin_amdgcn_make_buffer_rsrc((void*)p.W1, 0, FF_NUM_RECORDS, 0x00020000);
      const __amdgpu_buffer_rsrc_t rsW2 = __builtin_amdgcn_make_buffer_rsrc((void*)p.W2, 0, FF_NUM_RECORDS, 0x00020000);
      const int vo1 = (r * FF_D + 4 * g) * 4, vo2 = (r * F + 4 * g) * 4;
      auto wload = [&](int n0, int s) -> f32x4 {
        // GEMM2 column tiles are ROTATED by the wave: accumulator j of wave w holds column tile (j + 4 w) & 15, so "my own column
        // group" is registers 0..3 and "the group wave w + k owns" registers 4 k .. 4 k + 3 for EVERY wave -- the reduce-scatter
        // below never indexes registers by the wave id (which hipcc turns into a 1-KB scratch array)
        const int so = __builtin_amdgcn_readfirstlane(s < 16 ? (n0 * FF_D + s * 16) * 4 : ((((s - 16) + 4 * wave) & 15) * 16 * F + n0) * 4);
        const u32x4 v = s < 16 ? __builtin_amdgcn_raw_buffer_load_b128(rsW1, vo1, so, 0) : __builtin_amdgcn_raw_buffer_load_b128(rsW2, vo2, so, 0);
        return f32x4{__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3])};
      };
      // LDS fragment of row tile i, k-group kg: 16 B at k = 16 kg + 4 g of row 16 i + r
      // (the fragments are the same for every unit of the part: unless their offset LOOKS unit-dependent -- u * p.zero, zero at run
      //  time -- hipcc hoists all 64 ds_reads, 256 registers, out of the unit loop and spills the accumulators; an empty asm
      //  on the offset does it too, but draws an s_waitcnt vmcnt(0) in front of itself that drains the weight ring every unit)
      const int xoff0 = r * FF_XS + 4 * g;
      int xoff = xoff0;
      auto xfrag = [&](int i, int kg) -> f32x4 { return *reinterpret_cast<const f32x4*>(xs + xoff + i * 16 * FF_XS + kg * 16); };
      f32x4 ring[FF_RING];
      {
        const int n0 = my_u0 * FF_UN;
#pragma unroll
        for (int s = 0; s < FF_RING; ++s) ring[s] = wload(n0, s);
      }
      for (int u = 0; u < my_n; ++u) {
        const int n0 = (my_u0 + u) * FF_UN;
        const int n0_next = (u + 1 < my_n ? n0 + FF_UN : n0);      // after the last unit: a harmless re-read
        xoff = xoff0 + u * p.zero;
        const f32x4 bias1 = *reinterpret_cast<const f32x4*>(p.b1 + n0 + 4 * g);
        // GEMM1 is summed as every K = 256 linear of the pack-invariant routes (CANON_KBLOCK = 64, gemm.hpp): one chain per 4 k-groups,
        // block sums added in ascending order (round 6; one chain over 256 sat 1.4x farther from float64 than torch's CPU sgemm)
        f32x4 h[WMT], ht[WMT];
#pragma unroll
        for (int i = 0; i < WMT; ++i) h[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        f32x4 xa[WMT];
#pragma unroll
        for (int i = 0; i < WMT; ++i) xa[i] = xfrag(i, 0);
        // GEMM1: 16 k-groups x 16 MFMAs (4 accumulators, each revisited every 4th MFMA)
#pragma unroll
        for (int s = 0; s < 16; ++s) {
          const f32x4 wf = ring[s % FF_RING];
          ring[s % FF_RING] = wload(s + FF_RING < 32 ? n0 : n0_next, (s + FF_RING) & 31);
          f32x4 xb[WMT];
          if (s + 1 < 16) {
#pragma unroll
            for (int i = 0; i < WMT; ++i) xb[i] = xfrag(i, s + 1);
          }
#pragma unroll
          for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int i = 0; i < WMT; ++i) h[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[e], xa[i][e], h[i], 0, 0, 0);   // D = W1 . LN(x)^T
          if (s + 1 < 16) {
#pragma unroll
            for (int i = 0; i < WMT; ++i) xa[i] = xb[i];
          }
          if ((s & 3) == 3) {
#pragma unroll
            for (int i = 0; i < WMT; ++i) {
#pragma unroll
              for (int e = 0; e < 4; ++e) ht[i][e] = s == 3 ? h[i][e] : ht[i][e] + h[i][e];
              h[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
          }
          FF_STEP_FENCE;
        }
#pragma unroll
        for (int i = 0; i < WMT; ++i) h[i] = ht[i];
        // bias + SiLU in the accumulator layout (lane (r, g), register e = hidden column n0 + 4 g + e of row 16 i + r)
#pragma unroll
        for (int i = 0; i < WMT; ++i)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float v = h[i][e] + bias1[e];
            h[i][e] = v / (1.0f + __expf(-v));
          }
        // GEMM2: 16 column tiles x 16 MFMAs
#pragma unroll
        for (int s = 16; s < 32; ++s) {
          const f32x4 wf = ring[s % FF_RING];
          ring[s % FF_RING] = wload(s + FF_RING < 32 ? n0 : n0_next, (s + FF_RING) & 31);
#pragma unroll
          for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int i = 0; i < WMT; ++i)
              yacc[i][s - 16] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[e], h[i][e], yacc[i][s - 16], 0, 0, 0);   // D = W2 . H^T
          FF_STEP_FENCE;
        }
      }
    }

    // ---- the four waves' partial tiles -> wave w owns columns [64 w, 64 w + 64): 3-round reduce-scatter through LDS ----
    // round k: wave w parks its registers 4 k .. 4 k + 3 (column group (w + k) & 3), then adds what wave (w - k) & 3 parked
    // (that wave's registers 4 k .. are column group w) into its registers 0 .. 3: ((own + w-1) + w-2) + w-3, a fixed order
    __syncthreads();                                  // every wave is done reading xs
    f32x4* xr = reinterpret_cast<f32x4*>(xs);
#pragma unroll
    for (int k = 1; k < 4; ++k) {
#pragma unroll
      for (int i = 0; i < WMT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) xr[(wave * 4 * WMT + i * 4 + j) * 64 + lane] = yacc[i][4 * k + j];
      __syncthreads();
      const int src = (wave - k) & 3;
#pragma unroll
      for (int i = 0; i < WMT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const f32x4 o = xr[(src * 4 * WMT + i * 4 + j) * 64 + lane];
#pragma unroll
          for (int e = 0; e < 4; ++e) yacc[i][j][e] += o[e];
        }
      __syncthreads();
    }
    f32x4 fin[WMT][4];
#pragma unroll
    for (int i = 0; i < WMT; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) fin[i][j] = yacc[i][j];

    // ---- several workgroups share the tile: park, count, the last arrival finishes ----
    const int wf_ = (int)(((ut0 + 1) * p.G - 1) / U);             // workgroup that owns the tile's first unit
    const int wl_ = (int)(((ut0 + UT) * p.G - 1) / U);            // ... its last unit
    if (!p.canon && wl_ > wf_) {
      {
        const int slot = 2 * w + (tile == t_first ? 0 : 1);
        const __amdgpu_buffer_rsrc_t rsP = __builtin_amdgcn_make_buffer_rsrc((void*)(p.ws + (size_t)slot * FF_SLOT), 0, FF_SLOT * 4, 0x00020000);
#pragma unroll
        for (int i = 0; i < WMT; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            u32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = __float_as_uint(fin[i][j][e]);
            __builtin_amdgcn_raw_buffer_store_b128(v, rsP, ((wave * 4 * WMT + i * 4 + j) * 64 + lane) * 16, 0, FF_SC1);
          }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // EVERY storing wave drains its write-through stores
      __syncthreads();
      if (t == 0) s_misc[0] = (int)atomicAdd(p.cnt + tile, 1u);
      __syncthreads();
      const int arrived = __builtin_amdgcn_readfirstlane(s_misc[0]);
      if (arrived != wl_ - wf_) continue;                         // not the last: done with this tile
      if (t == 0) __hip_atomic_store(p.cnt + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // state for the next launch
      f32x4 tot[WMT][4];
#pragma unroll
      for (int i = 0; i < WMT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) tot[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      for (int ww = wf_; ww <= wl_; ++ww) {                        // fixed order: (((0 + P[wf]) + P[wf+1]) + ...)
        if (ww == w) {
#pragma unroll
          for (int i = 0; i < WMT; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
              for (int e = 0; e < 4; ++e) tot[i][j][e] += fin[i][j][e];
          continue;
        }
        const int tf = (int)(((long long)ww * U / p.G) / UT);     // the contributor's first tile
        const int slot = 2 * ww + (tile == tf ? 0 : 1);
        const __amdgpu_buffer_rsrc_t rsP = __builtin_amdgcn_make_buffer_rsrc((void*)(p.ws + (size_t)slot * FF_SLOT), 0, FF_SLOT * 4, 0x00020000);
        u32x4 o[WMT][4];
#pragma unroll
        for (int i = 0; i < WMT; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) o[i][j] = __builtin_amdgcn_raw_buffer_load_b128(rsP, ((wave * 4 * WMT + i * 4 + j) * 64 + lane) * 16, 0, FF_SC1);
#pragma unroll
        for (int i = 0; i < WMT; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) tot[i][j][e] += __uint_as_float(o[i][j][e]);
      }
#pragma unroll
      for (int i = 0; i < WMT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) fin[i][j] = tot[i][j];
    }

    // ---- epilogue: (sum + b2) * alpha + x  [-> LayerNorm over the row], columns [64 wave, 64 wave + 64) of rows m0 .. m0 + 63 ----
    const int col0 = wave * 64 + 4 * g;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const f32x4 b2v = *reinterpret_cast<const f32x4*>(p.b2 + col0 + j * 16);
#pragma unroll
      for (int i = 0; i < WMT; ++i) {
        const int m = min(m0 + i * 16 + r, p.M - 1);                // clamped: rows >= M are never stored
        const f32x4 xv = *reinterpret_cast<const f32x4*>(p.X + (size_t)m * p.ldx + col0 + j * 16);
#pragma unroll
        for (int e = 0; e < 4; ++e) fin[i][j][e] = (fin[i][j][e] + b2v[e]) * p.alpha + xv[e];
      }
    }
    if (p.ln2_g) {
      float mean[WMT], rstd[WMT];
#pragma unroll
      for (int i = 0; i < WMT; ++i) {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) s += (fin[i][j][0] + fin[i][j][1]) + (fin[i][j][2] + fin[i][j][3]);
        s += __shfl_xor(s, 16, 64);
        s += __shfl_xor(s, 32, 64);
        if (g == 0) red[(i * 16 + r) * 4 + wave] = s;
      }
      __syncthreads();
#pragma unroll
      for (int i = 0; i < WMT; ++i) {
        const float* q4 = red + (i * 16 + r) * 4;
        mean[i] = ((q4[0] + q4[1]) + (q4[2] + q4[3])) * (1.0f / FF_D);
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) { const float d = fin[i][j][e] - mean[i]; s += d * d; }
        s += __shfl_xor(s, 16, 64);
        s += __shfl_xor(s, 32, 64);
        if (g == 0) red[FF_BM * 4 + (i * 16 + r) * 4 + wave] = s;
      }
      __syncthreads();
#pragma unroll
      for (int i = 0; i < WMT; ++i) {
        const float* q4 = red + FF_BM * 4 + (i * 16 + r) * 4;
        rstd[i] = 1.0f / sqrtf(((q4[0] + q4[1]) + (q4[2] + q4[3])) * (1.0f / FF_D) + 1e-5f);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f32x4 gm = *reinterpret_cast<const f32x4*>(p.ln2_g + col0 + j * 16);
        const f32x4 bt = *reinterpret_cast<const f32x4*>(p.ln2_b + col0 + j * 16);
#pragma unroll
        for (int i = 0; i < WMT; ++i)
#pragma unroll
          for (int e = 0; e < 4; ++e) fin[i][j][e] = (fin[i][j][e] - mean[i]) * rstd[i] * gm[e] + bt[e];
      }
    }
#pragma unroll
    for (int i = 0; i < WMT; ++i) {
      const int m = m0 + i * 16 + r;
      if (m >= p.M) continue;
#pragma unroll
      for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(p.Y + (size_t)m * p.ldy + col0 + j * 16) = fin[i][j];
    }
  }
#endif
}

// ---- host side ---------------------------------------------------------------------------------
void ffn_fused_debug_grid(int g) { dispatch_edit([g](Dispatch& d) { d.ffn_force_g = g; }); }   // tests / tuning: fixed grid

// (16-row MFMA tiles per wave: Dispatch::ffn_wm, SS_FFN_WM; the pack-invariant form picks its own unless forced)
// wm 1..4: force that tile height; 0: back to the process default (SS_FFN_WM if set, else the heuristic); < 0: keep (0 used
// to clobber an SS_FFN_WM override with the built-in default)
void ffn_fused_debug_rows(int wm) {
  if (wm < 0) return;
  static const int env_wm = getenv("SS_FFN_WM") ? atoi(getenv("SS_FFN_WM")) : 0;
  dispatch_edit([wm](Dispatch& d) {
    if (wm >= 1 && wm <= 4) { d.ffn_wm = wm; d.ffn_wm_forced = 1; }
    else if (env_wm >= 1 && env_wm <= 4) { d.ffn_wm = env_wm; d.ffn_wm_forced = 1; }
    else { d.ffn_wm = 3; d.ffn_wm_forced = 0; }
  });
}

// relative cost of a row at tile height 16 h (h = 1..4), SS_FFN_COST="c1,c2,c3,c4" overrides (tuning)
struct FfnCost { double c[5]; };
static const FfnCost g_ffn_cost = [] {
  FfnCost k{{0.0, 1.88, 1.07, 1.0, 1.02}};     // us per 16 rows and round at 16- / 32- / 48- / 64-row tiles, relative (121 / 137 / 193 / 262 us per round: profiles/r05_ffn_canon_bench.txt)
  const char* e = getenv("SS_FFN_COST");
  if (e) sscanf(e, "%lf,%lf,%lf,%lf", &k.c[1], &k.c[2], &k.c[3], &k.c[4]);
  return k;
}();

// canon: the whole-tile form shares no tile between workgroups -- it needs no arrival counters, hence no bound on the row count.
// Otherwise the per-tile counters bound it: FF_MAX_TILES tiles at the LOWEST tile height a launch may take (16 rows: ss_debug_ffn /
// SS_FFN_WM allow it on this path; ADVICE r5: the bound used to assume 48-row tiles).
bool ffn_fused_eligible(int D, int F, int act, int M, int ldx, int ldy, bool canon) {
  return D == FF_D && F >= 64 && F % 64 == 0 && F <= 8192 && act == ACT_SILU && M > 0 && (ldx & 3) == 0 && (ldy & 3) == 0 &&
         (canon || (M + 15) / 16 <= FF_MAX_TILES);
}

template <int WMT>
static int launch_ffn_t(FfnKArgs a, int D, const float* ln2_g, hipStream_t stream) {
  constexpr int BM = 16 * WMT;
  constexpr size_t kLds = ff_lds_bytes(BM);
  SS_MAX_LDS_ONCE((&ffn_fused_kernel<WMT>), kLds);
  SkWorkspace* st = nullptr;
  int rc = sk_workspace_acquire(stream, &st);
  if (rc != SS_OK) return rc;
  const int tiles = cdiv(a.M, BM);
  const long long U = (long long)tiles * (a.F / FF_UN);
  if (!a.canon && tiles > FF_MAX_TILES) return SS_ERR_ARG;       // per-tile arrival counters (sync3) of this context
  if (a.canon) {
    // pack-invariant form: whole tiles per workgroup; two resident workgroups per CU at 16- / 32-row tiles (128 accumulator registers)
    long long G = disp().ffn_force_g > 0 ? disp().ffn_force_g : (long long)st->cus * (WMT <= 2 ? 2 : 1);
    if (G > tiles) G = tiles;
    if (G < 1) G = 1;
    a.G = (int)G; a.ws = st->ws; a.cnt = st->sync3; a.zero = 0;
    GemmArgs ga;
    ga.M = a.M; ga.N = D; ga.Cin = a.F; ga.in_len = a.M;
    ga.algo_flops = 4.0 * (double)a.M * D * a.F;
    ga.algo_bytes = 4.0 * (2.0 * (double)a.M * D + 2.0 * (double)D * a.F + a.F + 3.0 * D + (ln2_g ? 2.0 * D : 0.0));
    ProfRec rec{}; bool prof = false;
    rc = prof_begin(ga, stream, 22, rec, prof);
    if (rc != SS_OK) return rc;
    hipLaunchKernelGGL(ffn_fused_kernel<WMT>, dim3((unsigned)G), dim3(256), kLds, stream, a);
    SS_LAUNCH_CHECK();
    return prof_end(stream, rec, prof);
  }
  // One workgroup per CU.  A tile is shared by at most ~8 workgroups (each parks a partial that the tile's last arrival reads
  // back: beyond that the finisher's serial read is the kernel's tail), and every workgroup gets at least 4 units (one per wave).
  long long G = disp().ffn_force_g > 0 ? disp().ffn_force_g : st->cus;
  if (G > st->cus) G = st->cus;                       // slots: the context's workspace holds 2 x cus partials of 64 KB (two per workgroup)
  if (disp().ffn_force_g <= 0 && G > 8LL * tiles) G = 8LL * tiles;
  if (G > U / 4) G = U / 4;
  if (G < 1) G = 1;
  a.G = (int)G; a.ws = st->ws; a.cnt = st->sync3; a.zero = 0;
  GemmArgs ga;                                        // profiler class ffn_fused (22): both GEMMs; x in, y out, weights + biases + LN once
  ga.M = a.M; ga.N = D; ga.Cin = a.F; ga.in_len = a.M;
  ga.algo_flops = 4.0 * (double)a.M * D * a.F;
  ga.algo_bytes = 4.0 * (2.0 * (double)a.M * D + 2.0 * (double)D * a.F + a.F + 3.0 * D + (ln2_g ? 2.0 * D : 0.0));
  ProfRec rec{}; bool prof = false;
  rc = prof_begin(ga, stream, 22, rec, prof);
  if (rc != SS_OK) return rc;
  hipLaunchKernelGGL(ffn_fused_kernel<WMT>, dim3((unsigned)G), dim3(256), kLds, stream, a);
  SS_LAUNCH_CHECK();
  return prof_end(stream, rec, prof);
}

int launch_ffn_fused(const float* X, int ldx, float* Y, int ldy, const float* ln_g, const float* ln_b, const float* W1,
                     const float* b1, const float* W2, const float* b2, float alpha, const float* ln2_g, const float* ln2_b,
                     int M, int D, int F, hipStream_t stream, int canon) {
  if (!ffn_fused_eligible(D, F, ACT_SILU, M, ldx, ldy, canon != 0) || !X || !Y || !ln_g || !ln_b || !W1 || !b1 || !W2 || !b2) return SS_ERR_ARG;
  FfnKArgs a;
  a.X = X; a.ldx = ldx; a.Y = Y; a.ldy = ldy; a.ln_g = ln_g; a.ln_b = ln_b; a.W1 = W1; a.b1 = b1; a.W2 = W2; a.b2 = b2;
  a.ln2_g = ln2_g; a.ln2_b = ln2_g ? ln2_b : nullptr; a.alpha = alpha; a.M = M; a.F = F; a.canon = canon ? 1 : 0;
  int wm = disp().ffn_wm;
  if (canon && !disp().ffn_wm_forced) {
    // Tile height of the pack-invariant form (a row's bits do not depend on it): the one whose tiles go over the CUs in the fewest
    // rows per CU, weighted by what a row costs at that height (fewer MFMAs per weight fragment at low heights; tools/ffn_bench.py)
    SkWorkspace* st = nullptr;
    int rc = sk_workspace_acquire(stream, &st);
    if (rc != SS_OK) return rc;
    const double* cost = g_ffn_cost.c;
    double best = 1e300;
    for (int h = 1; h <= 4; ++h) {
      const long long tiles = cdiv(M, 16 * h);
      const double t = (double)((tiles + st->cus - 1) / st->cus) * 16 * h * cost[h];
      if (t < best) { best = t; wm = h; }
    }
  }
  switch (wm) {
    case 1: return launch_ffn_t<1>(a, D, ln2_g, stream);
    case 2: return launch_ffn_t<2>(a, D, ln2_g, stream);
    case 4: return launch_ffn_t<4>(a, D, ln2_g, stream);
    default: return launch_ffn_t<3>(a, D, ln2_g, stream);
  }
}

}  // namespace ss
