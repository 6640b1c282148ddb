// Row-tile linear layer for the K = 256 projections of packed batches (gfx950, exact-f32 MFMA):
//   C = epi( [LayerNorm](X) . W^T + b ),  epi = activation, * alpha, + R   or   GLU (value * sigmoid(gate))
// -- the encoder's QKV / attention-output / conv-module pointwise convs / input projection, both CTC vocabulary heads and the MT
// decoder's cross-attention K|V projection (reference chunk_unity/modules/conformer_layer.py:94-119, 254-312,
// uni_unity/modules/espnet_multihead_attention.py:41-59, fairseq/models/speech_to_speech/modules/ctc_decoder.py:11-18;
// SURVEY.md §8a rows a3-a8), at more rows than the no-LDS small-M kernel takes.
//
// Why not the LDS-tiled kernel: with K = 256 a 32 x 64 tile has 8 k-steps -- its prologue, epilogue and the
// L2 -> LDS staging of BOTH operands bound it at 0.36-0.46 of the FP32-MFMA peak.  The structure that works for the fused FFN
// (ffn.hip) is used here for a single GEMM:
//   * the (48-row tile, 16-column unit) space of a launch is cut into equal contiguous ranges, one per workgroup (one per CU);
//     a part's row tile goes ONCE into LDS ([48][260] floats, LayerNorm applied on the way when the layer has one) and is read
//     by the four waves (one per SIMD), which take disjoint quarters of the part's units;
//   * weight fragments go from L2 straight to registers: one buffer load of 16 B per lane and k-group with the (unit, k-group)
//     part of the address in the wave-uniform soffset, a ring of 8 requested 4-8 steps ahead; a W fragment feeds 12 MFMAs, an
//     LDS fragment 4; no barrier, no LDS-DMA piece and no LDS write inside the contraction;
//   * swapped MFMA operands (D = W . X^T): a lane holds 4 consecutive output channels of one row, so bias / residual / output
//     move as float4; a unit (192 MFMAs) ends in its own epilogue -- units are independent, nothing is reduced across waves or
//     workgroups, results do not depend on the grid.
// GLU producers keep the pack-time [16 value | 16 gate] row interleave: a unit is then 32 weight rows -> 16 output columns.
#include "gemm.hpp"

#include <cstdlib>

#ifndef RT_FENCE
#define RT_FENCE 1      // see ffn.hip: without a fence per step hipcc sinks every weight load to just before its use
#endif
#if RT_FENCE
#define RT_STEP_FENCE __builtin_amdgcn_sched_barrier(0)
#else
#define RT_STEP_FENCE do { } while (0)
#endif

namespace ss {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

namespace {

constexpr int RT_K = 256;                      // contraction length (enc_dim)
constexpr int RT_WM = 3;                       // 16-row MFMA tiles per wave
constexpr int RT_BM = 16 * RT_WM;              // rows per tile
constexpr int RT_XS = RT_K + 4;                // LDS row stride (floats): 260 = 4 mod 64 banks, fragment addresses = one base + immediates
[[maybe_unused]] constexpr int RT_RING = 8;
[[maybe_unused]] constexpr int RT_NUM_RECORDS = 0x7ffffff0;
constexpr size_t RT_LDS = (size_t)RT_BM * RT_XS * sizeof(float);

struct RtArgs {
  const float* X; int ldx;
  const float* W; const float* bias;
  const float *ln_g, *ln_b;                    // LayerNorm over the K input columns, or null
  const float* R; int ldr;                     // residual added after alpha, or null
  float* C; int ldc;
  float alpha; int act;
  int M, NU, G;                                // rows, units per tile (N / 16, GLU: N / 32), workgroups
  int zero;                                    // 0 at run time, opaque at compile time (see xoff)
};

}  // namespace

template <bool GLU>
__global__ __launch_bounds__(256, GLU ? 2 : 3) void rt_linear_kernel(const RtArgs p) {     // (GLU: two accumulator + two total sets -- 3 per CU would spill)
#if __HIP_DEVICE_COMPILE__
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* xs = smem;
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int r = lane & 15, g = lane >> 4;
  const int w = blockIdx.x;
  const int NU = p.NU;
  const int tiles = (p.M + RT_BM - 1) / RT_BM;
  const long long U = (long long)tiles * NU;
  const long long u0 = (long long)w * U / p.G, u1 = (long long)(w + 1) * U / p.G;
  if (u1 <= u0) return;
  const int t_first = (int)(u0 / NU), t_last = (int)((u1 - 1) / NU);
  constexpr int WROWS = GLU ? 32 : 16;          // weight rows per unit
  constexpr int NLD = GLU ? 32 : 16;            // weight fragments per unit (k-group major; GLU: value, gate per k-group)

  const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, RT_NUM_RECORDS, 0x00020000);
  const int vo = (r * RT_K + 4 * g) * 4;
  // fragment f of the unit whose first weight row is n0: k-group f (GLU: k-group f / 2 of the value (f even) / gate (f odd) block)
  auto wload = [&](int n0, int f) -> f32x4 {
    const int so = __builtin_amdgcn_readfirstlane(GLU ? ((n0 + (f & 1) * 16) * RT_K + (f >> 1) * 16) * 4 : (n0 * RT_K + f * 16) * 4);
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsW, vo, so, 0);
    return f32x4{__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3])};
  };

  for (int tile = t_first; tile <= t_last; ++tile) {
    const long long ut0 = (long long)tile * NU;
    const int ka = (int)(max(u0, ut0) - ut0), kb = (int)(min(u1, ut0 + NU) - ut0);
    const int m0 = tile * RT_BM;

    // ---- the row tile into LDS, LayerNorm on the way (thread = row t / 4, 16-B chunks q, q + 4, ...) ----
    __syncthreads();                                  // every wave is done reading the previous part's tile
    if (t < 4 * RT_BM) {
      const int row = t >> 2, q = t & 3;
      const int m = min(m0 + row, p.M - 1);           // clamped (branch-free loads): rows >= M are computed on a copy, never stored
      f32x4 v[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) v[i] = *reinterpret_cast<const f32x4*>(p.X + (size_t)m * p.ldx + (i * 4 + q) * 4);
      if (p.ln_g) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
        s += __shfl_xor(s, 1, 64);
        s += __shfl_xor(s, 2, 64);
        const float mean = s * (1.0f / RT_K);
        float qq = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i)
#pragma unroll
          for (int e = 0; e < 4; ++e) { const float d = v[i][e] - mean; qq += d * d; }
        qq += __shfl_xor(qq, 1, 64);
        qq += __shfl_xor(qq, 2, 64);
        const float rstd = 1.0f / sqrtf(qq * (1.0f / RT_K) + 1e-5f);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int c = i * 4 + q;
          const f32x4 gm = *reinterpret_cast<const f32x4*>(p.ln_g + c * 4);
          const f32x4 bt = *reinterpret_cast<const f32x4*>(p.ln_b + c * 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[i][e] = (v[i][e] - mean) * rstd * gm[e] + bt[e];
        }
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) *reinterpret_cast<f32x4*>(xs + row * RT_XS + ((i * 4 + q) << 2)) = v[i];
    }
    __syncthreads();

    // ---- this wave's units of the part: a contiguous quarter of [ka, kb) ----
    const int n_part = kb - ka;
    const int ubase = n_part >> 2, urem = n_part & 3;
    const int my_n = ubase + (wave < urem ? 1 : 0);
    const int my_u0 = ka + wave * ubase + (wave < urem ? wave : urem);
    if (my_n <= 0) continue;

    // (the LDS fragments are the same for every unit: their offset must LOOK unit-dependent or hipcc hoists all 48 reads)
    const int xoff0 = r * RT_XS + 4 * g;
    int xoff = xoff0;
    auto xfrag = [&](int i, int kg) -> f32x4 { return *reinterpret_cast<const f32x4*>(xs + xoff + i * 16 * RT_XS + kg * 16); };
    f32x4 ring[RT_RING];
#pragma unroll
    for (int f = 0; f < RT_RING; ++f) ring[f] = wload(my_u0 * WROWS, f);
    for (int u = 0; u < my_n; ++u) {
      const int unit = my_u0 + u;
      const int n0 = unit * WROWS;
      const int n0_next = (u + 1 < my_n ? n0 + WROWS : n0);        // after the last unit: a harmless re-read
      xoff = xoff0 + u * p.zero;
      // bias and residual of the unit are requested BEFORE its contraction: vmcnt counts in order, so an epilogue that loaded
      // them at its start would wait for the whole weight ring (the newest fragments are one step old) at the end of every unit
      const int oc = unit * 16 + 4 * g;
      f32x4 bb = f32x4{0.f, 0.f, 0.f, 0.f}, bg = f32x4{0.f, 0.f, 0.f, 0.f}, rr[RT_WM];
      if (p.bias) {
        bb = *reinterpret_cast<const f32x4*>(p.bias + (GLU ? n0 + 4 * g : oc));
        if constexpr (GLU) bg = *reinterpret_cast<const f32x4*>(p.bias + n0 + 16 + 4 * g);
      }
#pragma unroll
      for (int i = 0; i < RT_WM; ++i) {
        rr[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (!GLU && p.R) rr[i] = *reinterpret_cast<const f32x4*>(p.R + (size_t)min(m0 + i * 16 + r, p.M - 1) * p.ldr + oc);
      }
      // CANON_KBLOCK = 64 (gemm.hpp): one chain per 4 slabs, block sums added in ascending order into tot / totg
      f32x4 acc[RT_WM], accg[GLU ? RT_WM : 1], tot[RT_WM], totg[GLU ? RT_WM : 1];
#pragma unroll
      for (int i = 0; i < RT_WM; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      if constexpr (GLU) {
#pragma unroll
        for (int i = 0; i < RT_WM; ++i) accg[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      f32x4 xa[RT_WM];
#pragma unroll
      for (int i = 0; i < RT_WM; ++i) xa[i] = xfrag(i, 0);
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        f32x4 xb[RT_WM];
        if (s + 1 < 16) {
#pragma unroll
          for (int i = 0; i < RT_WM; ++i) xb[i] = xfrag(i, s + 1);
        }
        if constexpr (GLU) {
          const f32x4 wv = ring[(2 * s) % RT_RING], wg = ring[(2 * s + 1) % RT_RING];
          ring[(2 * s) % RT_RING] = wload(2 * s + RT_RING < NLD ? n0 : n0_next, (2 * s + RT_RING) % NLD);
          ring[(2 * s + 1) % RT_RING] = wload(2 * s + 1 + RT_RING < NLD ? n0 : n0_next, (2 * s + 1 + RT_RING) % NLD);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
#pragma unroll
            for (int i = 0; i < RT_WM; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[e], xa[i][e], acc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < RT_WM; ++i) accg[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(wg[e], xa[i][e], accg[i], 0, 0, 0);
          }
        } else {
          const f32x4 wf = ring[s % RT_RING];
          ring[s % RT_RING] = wload(s + RT_RING < NLD ? n0 : n0_next, (s + RT_RING) % NLD);
#pragma unroll
          for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int i = 0; i < RT_WM; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[e], xa[i][e], acc[i], 0, 0, 0);   // D = W . X^T
        }
        if (s + 1 < 16) {
#pragma unroll
          for (int i = 0; i < RT_WM; ++i) xa[i] = xb[i];
        }
        if ((s & 3) == 3) {                            // end of a 64-wide k-block
#pragma unroll
          for (int i = 0; i < RT_WM; ++i) {
#pragma unroll
            for (int e = 0; e < 4; ++e) tot[i][e] = s == 3 ? acc[i][e] : tot[i][e] + acc[i][e];
            acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if constexpr (GLU) {
#pragma unroll
              for (int e = 0; e < 4; ++e) totg[i][e] = s == 3 ? accg[i][e] : totg[i][e] + accg[i][e];
              accg[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
          }
        }
        RT_STEP_FENCE;
      }
#pragma unroll
      for (int i = 0; i < RT_WM; ++i) {
        acc[i] = tot[i];
        if constexpr (GLU) accg[i] = totg[i];
      }
      // ---- epilogue of the unit: lane (r, g) holds output columns oc .. oc + 3 of rows m0 + 16 i + r ----
      if constexpr (GLU) {
#pragma unroll
        for (int i = 0; i < RT_WM; ++i) {
          const int m = m0 + i * 16 + r;
          f32x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float val = acc[i][e] + bb[e], gate = accg[i][e] + bg[e];
            o[e] = val * (1.0f / (1.0f + expf(-gate)));
          }
          if (m < p.M) *reinterpret_cast<f32x4*>(p.C + (size_t)m * p.ldc + oc) = o;
        }
      } else {
#pragma unroll
        for (int i = 0; i < RT_WM; ++i) {
          const int m = m0 + i * 16 + r;
          f32x4 v = acc[i];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += bb[e];
          if (p.act == ACT_SILU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = v[e] / (1.0f + expf(-v[e]));
          } else if (p.act == ACT_RELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] *= p.alpha;
          if (p.R) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += rr[i][e];
          }
          if (m < p.M) *reinterpret_cast<f32x4*>(p.C + (size_t)m * p.ldc + oc) = v;
        }
      }
    }
  }
#endif
}


// ================================================================================================================================
// K > 256 (round 6): the decoder-side linears of a packed batch -- T2U encoder / unit decoder QKV (512 -> 1536), attention outputs
// and cross projections (512 -> 512 / 1024), FFN halves (512 -> 2048 with ReLU, 2048 -> 512 + residual) at 45-60 k rows
// (researches/ctc_unity/modules/transformer_layer.py:388-551, fairseq/models/speech_to_speech/modules/ctc_decoder.py:11-18;
// SURVEY.md §8a rows a11-a12).  Rounds 4-5 ran them on conv_sk2 cut on whole tiles: ONE accumulator chain over K <= 2048 (2.8x
// farther from float64 than torch's blocked sgemm at K = 2048) at 3x the algorithmic HBM traffic (both operands
// staged through LDS per 256 x 128 tile).  This kernel is the row-tile structure above with the row tile going through LDS one
// 256-wide k-slice at a time, and the pack-invariant summation of round 6 (CANON_KBLOCK = 64: one chain inside a 64-wide block,
// block sums added in ascending order -- bit-identical to conv_gemm_kernel<.., BLK = true> whatever the row count):
//   * work unit = (48-row tile, column group of 4 waves x UW x 16 columns); the (tile, group) space is cut into equal contiguous
//     ranges, one per workgroup; the groups of a tile are adjacent, so a tile's A rows are re-read from L2;
//   * per k-slice the tile's [48 x 256] part goes once into LDS; each wave contracts its UW 16-column units against it
//     (192 MFMAs per unit and slice, weight fragments L2 -> registers through the ring of 8), every 64 k the accumulator is
//     added to the unit's running total (UW x 12 registers) and cleared;
//   * epilogue per unit after the last block: bias / activation / alpha / residual as float4.
// ================================================================================================================================
namespace {
struct RtKbArgs {
  const float* X; int ldx;
  const float* W; const float* bias;
  const float* R; int ldr;
  float* C; int ldc;
  float alpha; int act;
  int M, K, NCG, G;                            // rows, contraction length (multiple of 256), column groups per tile, workgroups
  int zero;
  int xmap;                                    // != 0: XCD-aware unit order (see the kernel's unit loop); needs G % 8 == 0 and (G / 8) % NCG == 0
};
}  // namespace

template <int UW>
__global__ __launch_bounds__(256, 3) void rt_linear_kb_kernel(const RtKbArgs p) {
#if __HIP_DEVICE_COMPILE__
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* xs = smem;
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int r = lane & 15, g = lane >> 4;
  const int w = blockIdx.x;
  const int K = p.K, KB = K / RT_K;
  const int tiles = (p.M + RT_BM - 1) / RT_BM;
  const long long U = (long long)tiles * p.NCG;
  const long long u0 = (long long)w * U / p.G, u1 = (long long)(w + 1) * U / p.G;
  constexpr int CG = 4 * UW * 16;               // columns per group

  const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, RT_NUM_RECORDS, 0x00020000);
  const int vo = (r * K + 4 * g) * 4;
  auto wload = [&](int n0, int kb, int f) -> f32x4 {      // fragment f (k-group) of block kb of the unit whose first weight row is n0
    const int so = __builtin_amdgcn_readfirstlane((n0 * K + kb * RT_K + f * 16) * 4);
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsW, vo, so, 0);
    return f32x4{__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3])};
  };
  const int xoff0 = r * RT_XS + 4 * g;

  // Unit order.  Plain: a contiguous range of (tile, column group) units per workgroup -- a workgroup then walks the column groups of ONE
  // row tile one after the other, and by the time it comes back for the tile's A rows the XCD's L2 (4 MB = one weight matrix of this
  // size) has turned over: the A slice came from HBM N / 256 times (PMC: 2.7x the algorithmic bytes).  XCD-aware (p.xmap): workgroup w
  // sits on XCD w % 8 (observed placement; only speed depends on it) with local index q = w / 8; in step s the XCD's G / 8 workgroups
  // take G / 8 / NCG row tiles and ALL their column groups side by side (tile = (s tps + q / NCG) 8 + x, group = q % NCG), so a tile's
  // k-blocks are fetched once and hit in L2 for the other groups.  Every unit is still computed whole by one workgroup: same bits.
  const int xx = w & 7, xq = w >> 3;
  const int tps = p.xmap ? (p.G >> 3) / p.NCG : 1;
  long long nit = u1 - u0;
  if (p.xmap) { const int first = (xq / p.NCG) * 8 + xx; nit = first < tiles ? (tiles - first + tps * 8 - 1) / (tps * 8) : 0; }
  for (long long it = 0; it < nit; ++it) {
    int tile, cg;
    if (p.xmap) { tile = ((int)it * tps + xq / p.NCG) * 8 + xx; cg = xq % p.NCG; }
    else { const long long unit = u0 + it; tile = (int)(unit / p.NCG); cg = (int)(unit - (long long)tile * p.NCG); }
    const int m0 = tile * RT_BM;
    const int nw0 = cg * CG + wave * (UW * 16);              // this wave's first column of the group
    f32x4 tot[UW][RT_WM];
    f32x4 ring[RT_RING];
#pragma unroll
    for (int f = 0; f < RT_RING; ++f) ring[f] = wload(nw0, 0, f);

    for (int kb = 0; kb < KB; ++kb) {
      // ---- the tile's k-block into LDS (thread = row t / 4, 16-B chunks q, q + 4, ...) ----
      __syncthreads();                                  // every wave is done reading the previous block
      if (t < 4 * RT_BM) {
        const int row = t >> 2, q = t & 3;
        const int m = min(m0 + row, p.M - 1);           // clamped: rows >= M are computed on a copy, never stored
        const float* src = p.X + (size_t)m * p.ldx + kb * RT_K;
        f32x4 v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = *reinterpret_cast<const f32x4*>(src + (i * 4 + q) * 4);
#pragma unroll
        for (int i = 0; i < 16; ++i) *reinterpret_cast<f32x4*>(xs + row * RT_XS + ((i * 4 + q) << 2)) = v[i];
      }
      __syncthreads();
      const int kb_next = (kb + 1 < KB ? kb + 1 : kb);  // after the last block: a harmless re-read
      int xoff = xoff0;
      auto xfrag = [&](int i, int kg) -> f32x4 { return *reinterpret_cast<const f32x4*>(xs + xoff + i * 16 * RT_XS + kg * 16); };
#pragma unroll
      for (int u = 0; u < UW; ++u) {
        const int n0 = nw0 + u * 16;
        const int n_nx = (u + 1 < UW ? n0 + 16 : nw0), kb_nx = (u + 1 < UW ? kb : kb_next);
        xoff = xoff0 + (u + kb) * p.zero;               // (the fragments are the same for every unit: must LOOK unit-dependent, see above)
        f32x4 acc[RT_WM];
#pragma unroll
        for (int i = 0; i < RT_WM; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        f32x4 xa[RT_WM];
#pragma unroll
        for (int i = 0; i < RT_WM; ++i) xa[i] = xfrag(i, 0);
#pragma unroll
        for (int s = 0; s < 16; ++s) {
          f32x4 xb[RT_WM];
          if (s + 1 < 16) {
#pragma unroll
            for (int i = 0; i < RT_WM; ++i) xb[i] = xfrag(i, s + 1);
          }
          const f32x4 wf = ring[s % RT_RING];
          ring[s % RT_RING] = s + RT_RING < 16 ? wload(n0, kb, s + RT_RING) : wload(n_nx, kb_nx, s + RT_RING - 16);
#pragma unroll
          for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int i = 0; i < RT_WM; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[e], xa[i][e], acc[i], 0, 0, 0);   // D = W . X^T
          if (s + 1 < 16) {
#pragma unroll
            for (int i = 0; i < RT_WM; ++i) xa[i] = xb[i];
          }
          if ((s & 3) == 3) {                          // end of a 64-wide k-block: block sum -> running total (ascending; the first block IS the total)
#pragma unroll
            for (int i = 0; i < RT_WM; ++i) {
#pragma unroll
              for (int e = 0; e < 4; ++e) tot[u][i][e] = (kb == 0 && s == 3) ? acc[i][e] : tot[u][i][e] + acc[i][e];
              acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
          }
          RT_STEP_FENCE;
        }
      }
    }

    // ---- epilogue: lane (r, g) holds output columns oc .. oc + 3 of rows m0 + 16 i + r of every unit ----
#pragma unroll
    for (int u = 0; u < UW; ++u) {
      const int oc = nw0 + u * 16 + 4 * g;
      f32x4 bb = f32x4{0.f, 0.f, 0.f, 0.f};
      if (p.bias) bb = *reinterpret_cast<const f32x4*>(p.bias + oc);
#pragma unroll
      for (int i = 0; i < RT_WM; ++i) {
        const int m = m0 + i * 16 + r;
        f32x4 v = tot[u][i];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += bb[e];
        if (p.act == ACT_SILU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = v[e] / (1.0f + expf(-v[e]));
        } else if (p.act == ACT_RELU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] *= p.alpha;
        if (m < p.M) {
          if (p.R) {
            const f32x4 rr = *reinterpret_cast<const f32x4*>(p.R + (size_t)m * p.ldr + oc);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += rr[e];
          }
          *reinterpret_cast<f32x4*>(p.C + (size_t)m * p.ldc + oc) = v;
        }
      }
    }
  }
#endif
}

// ---- host side ---------------------------------------------------------------------------------
// Smallest launch, in 192-MFMA units (a GLU unit counts twice), that goes to this kernel: every workgroup pays a row-tile prologue
// (48 KB into LDS, ~2.5 us), so with few units per tile and workgroup -- N = 256 at any row count, N = 512 / 768 below ~4000 rows --
// the 32 x 32 / 32 x 64 tiles win (profiles/r04_rtlin_bench.txt: attn_out 4200 rows 11.7 vs 14.8 us; qkv 33.1 vs 25.9; CTC head
// 156 vs 109 us = 0.75 of the FP32-MFMA peak).
// (Dispatch::rt_min_units, SS_RTLIN_MIN_UNITS; SS_NO_RTLIN / SS_RTLIN_MIN_ROWS: the A/B knobs -- dispatch.hpp)
void rtlin_debug(int grid, int enable) { dispatch_edit([grid, enable](Dispatch& d) { d.rt_force_g = grid; if (enable >= 0) d.rt_off = enable ? 0 : 1; }); }

bool rtlin_shape_ok(const GemmArgs& a) {
  const int M = a.M;
  const bool glu_ok = !a.glu || (a.N % 32 == 0 && a.act == ACT_NONE && a.alpha == 1.f && !a.R);
  return a.taps == 1 && a.stride == 1 && a.pad == 0 && a.Cin == RT_K && a.chunk == 0 && a.in_act == ACT_NONE && !a.R2 && !a.C2 &&
         a.div == 0.f && !a.ln_out && !a.x3 && glu_ok && a.N % 16 == 0 && a.N >= 16 && M >= 1 && a.nseg == 0 &&
         (a.act == ACT_NONE || a.act == ACT_SILU || a.act == ACT_RELU) && (a.lda & 3) == 0 && (a.ldc & 3) == 0 &&
         (!a.R || (a.ldr & 3) == 0) && a.A != a.C && a.in_len >= M && (!a.ln_g || a.ln_b) &&
         (size_t)a.N * RT_K * 4 < 0x7ff00000ull;
}

bool rtlin_eligible(const GemmArgs& a) {
  if (disp().rt_off) return false;
  const int M = a.M;
  return rtlin_shape_ok(a) && M >= disp().rt_min_rows &&
         (disp().rt_force_g > 0 || a.N >= 2048 ||                      // >= 128 units per row tile: the prologue is noise at any row count
          (long long)cdiv(M, RT_BM) * (a.N / 16) >= disp().rt_min_units);      // (N / 16: a GLU unit is 32 weight rows)
}

int launch_rtlin(const GemmArgs& a, hipStream_t stream) {
  if (!rtlin_shape_ok(a)) return SS_ERR_ARG;
  int cus = 0;
  {
    SkWorkspace* st = nullptr;                     // (only for the device's CU count, cached per context)
    int rc = sk_workspace_acquire(stream, &st);
    if (rc != SS_OK) return rc;
    cus = st->cus;
  }
  RtArgs q;
  q.X = a.A; q.ldx = a.lda; q.W = a.W; q.bias = a.bias; q.ln_g = a.ln_g; q.ln_b = a.ln_b; q.R = a.R; q.ldr = a.ldr; q.C = a.C; q.ldc = a.ldc;
  q.alpha = a.alpha; q.act = a.act; q.M = a.M; q.NU = a.glu ? a.N / 32 : a.N / 16; q.zero = 0;
  const long long U = (long long)cdiv(a.M, RT_BM) * q.NU;
  // 134 registers and 49 KB of LDS per workgroup: up to three are resident per CU (waves of different workgroups share a SIMD, one's
  // tile prologue / unit epilogues run under the other's MFMAs); units are independent, so any grid gives the same bits
  // (measured: one per CU unless a workgroup would hold >= 64 units -- the vocabulary heads -- where three per CU gain 3-9 %)
  static const int per_cu_env = getenv("SS_RTLIN_WG_PER_CU") ? atoi(getenv("SS_RTLIN_WG_PER_CU")) : 0;
  const int per_cu = min(a.glu ? 2 : 3, per_cu_env > 0 ? per_cu_env : (U >= 64LL * cus ? 3 : 1));
  long long G = disp().rt_force_g > 0 ? disp().rt_force_g : (long long)cus * per_cu;      // at least 4 units (one per wave) each
  if (disp().rt_force_g <= 0 && G > U / 4) G = U / 4;
  if (G > U) G = U;
  if (G < 1) G = 1;
  q.G = (int)G;
  ProfRec rec{}; bool prof = false;
  int rc = prof_begin(a, stream, 23, rec, prof);
  if (rc != SS_OK) return rc;
  if (a.glu) {
    SS_MAX_LDS_ONCE((&rt_linear_kernel<true>), RT_LDS);
    hipLaunchKernelGGL(rt_linear_kernel<true>, dim3((unsigned)G), dim3(256), RT_LDS, stream, q);
  } else {
    SS_MAX_LDS_ONCE((&rt_linear_kernel<false>), RT_LDS);
    hipLaunchKernelGGL(rt_linear_kernel<false>, dim3((unsigned)G), dim3(256), RT_LDS, stream, q);
  }
  SS_LAUNCH_CHECK();
  return prof_end(stream, rec, prof);
}


// ---- K > 256 (rt_linear_kb_kernel) ----
bool rtlin_kb_shape_ok(const GemmArgs& a) {
  return a.taps == 1 && a.stride == 1 && a.pad == 0 && a.Cin > RT_K && a.Cin % RT_K == 0 && a.Cin <= 8192 && a.chunk == 0 &&
         a.in_act == ACT_NONE && !a.R2 && !a.C2 && a.div == 0.f && !a.ln_out && !a.ln_g && !a.x3 && !a.glu && a.N % 256 == 0 && a.M >= 1 &&
         a.nseg == 0 && (a.act == ACT_NONE || a.act == ACT_SILU || a.act == ACT_RELU) && (a.lda & 3) == 0 && (a.ldc & 3) == 0 &&
         (!a.R || (a.ldr & 3) == 0) && a.A != a.C && a.in_len >= a.M && (size_t)a.N * a.Cin * 4 < 0x7ff00000ull;
}

bool rtlin_kb_eligible(const GemmArgs& a) {
  if (disp().rt_off || !rtlin_kb_shape_ok(a)) return false;
  // worth it from about one (tile, 64-column group) unit per CU (profiles/r06_rtlin_kb_bench.txt: 2500 rows x 512 columns, K = 2048:
  // 60 vs 72 us on the 32 x 64 tiles)
  return disp().rt_force_g > 0 || (long long)cdiv(a.M, RT_BM) * (a.N / 64) >= disp().rt_kb_min_units;
}

int launch_rtlin_kb(const GemmArgs& a, hipStream_t stream) {
  if (!rtlin_kb_shape_ok(a)) return SS_ERR_ARG;
  int cus = 0;
  {
    SkWorkspace* st = nullptr;
    int rc = sk_workspace_acquire(stream, &st);
    if (rc != SS_OK) return rc;
    cus = st->cus;
  }
  RtKbArgs q;
  q.X = a.A; q.ldx = a.lda; q.W = a.W; q.bias = a.bias; q.R = a.R; q.ldr = a.ldr; q.C = a.C; q.ldc = a.ldc;
  q.alpha = a.alpha; q.act = a.act; q.M = a.M; q.K = a.Cin; q.zero = 0;
  // 16-column units per wave: 4 (256-column groups: the tile's A slice is re-read N / 256 times, from L2) unless the launch then has
  // fewer units than three workgroups per CU -- short packs halve the group width until it has (profiles/r06_rtlin_kb_bench.txt:
  // 9000 rows x 512 columns, K = 2048: 376 units on 768 workgroup slots at UW = 4)
  int uw = disp().rt_kb_uw == 1 || disp().rt_kb_uw == 2 ? disp().rt_kb_uw : 4;
  if (disp().rt_kb_uw <= 0)
    while (uw > 1 && (long long)cdiv(a.M, RT_BM) * (a.N / (64 * uw)) < 3LL * cus) uw >>= 1;
  q.NCG = a.N / (64 * uw);
  const long long U = (long long)cdiv(a.M, RT_BM) * q.NCG;
  long long G = disp().rt_force_g > 0 ? disp().rt_force_g : 3LL * cus;
  if (G > U) G = U;
  if (G < 1) G = 1;
  q.G = (int)G;
  q.xmap = (disp().rt_kb_xmap && disp().rt_force_g <= 0 && G == 3LL * cus && (cus & 7) == 0 && ((G >> 3) % q.NCG) == 0 && U >= 2 * G) ? 1 : 0;
  ProfRec rec{}; bool prof = false;
  int rc = prof_begin(a, stream, 31, rec, prof);
  if (rc != SS_OK) return rc;
  if (uw == 1) {
    SS_MAX_LDS_ONCE((&rt_linear_kb_kernel<1>), RT_LDS);
    hipLaunchKernelGGL(rt_linear_kb_kernel<1>, dim3((unsigned)G), dim3(256), RT_LDS, stream, q);
  } else if (uw == 2) {
    SS_MAX_LDS_ONCE((&rt_linear_kb_kernel<2>), RT_LDS);
    hipLaunchKernelGGL(rt_linear_kb_kernel<2>, dim3((unsigned)G), dim3(256), RT_LDS, stream, q);
  } else {
    SS_MAX_LDS_ONCE((&rt_linear_kb_kernel<4>), RT_LDS);
    hipLaunchKernelGGL(rt_linear_kb_kernel<4>, dim3((unsigned)G), dim3(256), RT_LDS, stream, q);
  }
  SS_LAUNCH_CHECK();
  return prof_end(stream, rec, prof);
}

}  // namespace ss
