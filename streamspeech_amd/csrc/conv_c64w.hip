// Winograd F(2,3) form of the 64-channel HiFi-GAN stage convs of a packed batch (the launches conv_c64.hip takes: C = N = 64, k = 3 / 7 /
// 11, dilation 1 / 3 / 5; reference fairseq/models/text_to_speech/hifigan.py:52-172).
//
// A k-tap dilated conv is ceil(k / 3) groups of three taps (the last group zero-padded).  On the dilation lattice a group is a
// minimal-filtering problem F(2,3): the outputs of rows t and t + d need the inputs u, u + d, u + 2 d, u + 3 d (u = t - pad + 3 g d) and
// FOUR channel-mixing products instead of six,
//     m0 = G0 (x0 - x2)    m1 = G1 (x1 + x2)    m2 = G2 (x2 - x1)    m3 = G3 (x1 - x3)        y[t] = m0 + m1 + m2    y[t + d] = m1 - m2 - m3
//     G0 = w0,  G1 = (w0 + w1 + w2) / 2,  G2 = (w0 - w1 + w2) / 2,  G3 = w2      (64 x 64 matrices, transformed once per layer: wino_pack)
// and since the output transform is linear the four products are ACCUMULATED over the groups and the input channels in four
// accumulator sets, transformed once at the end: 4 G MFMA k-blocks per output pair instead of 2 k -- 1.5x fewer matrix-core cycles at
// k = 3, 1.17x at k = 7, 1.375x at k = 11.  In float32 the result is as close to float64 as the direct conv's (1.9e-7 relative either
// way: tools/winograd_error.py, profiles/r04_winograd_error.txt), so every parity bar of the direct kernels holds unchanged.
//
// Structure = conv_c64.hip (slab of the block's input rows once into LDS with the input leaky-ReLU applied, weight fragments from L2
// straight into registers through a ring, two workgroups per CU, swapped MFMA operands) with
//   * a wave tile of 32 output PAIRS x 64 columns x 4 transform components = 32 accumulator tiles (128 registers);
//   * the input transform on the A-fragment path: a sub-step (group, channel block, component) reads the two slab rows of its
//     component as two ds_read_b128 per pair tile and combines them with one VALU op per element -- per-lane LDS addresses make the
//     row pairing free (pair p of a block = rows 2 d (p / d) + p % d and + d), which is why this form fits the slab kernels and not
//     conv_sk2 (its A tiles arrive by LDS-DMA);
//   * 4 weight fragments + 4 LDS reads + 8 VALU ops per 32 MFMAs, ring of 8 fragments = two sub-steps (2048 MFMA cycles) ahead.
// Blocks cover 256 / 252 / 240 output rows at dilation 1 / 3 / 5 (whole pairs; two slabs must fit a CU's LDS); a conv whose slab
// does not fit two per CU (k = 11 at dilation 5) stays on conv_c64.hip.
#include "gemm.hpp"

#include <cstdlib>
#include <type_traits>

namespace ss {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

// Diagnostic build only (-DCW_TIMING=1, tools/cw_timing.py): thread 0 of every workgroup accumulates s_memtime cycles per phase.
#ifndef CW_TIMING
#define CW_TIMING 0
#endif
#if CW_TIMING
__device__ unsigned long long g_cw_dbg[1024 * 8];
#define CW_STAMP(k) do { if (t == 0) { const unsigned long long _n = __builtin_amdgcn_s_memtime(); cw_acc[k] += _n - cw_last; cw_last = _n; } } while (0)
#else
#define CW_STAMP(k) do { } while (0)
#endif

namespace {
constexpr int CW_MAXSEG = 256;
constexpr int CW_MAXROWS = 320;                // slab rows a thread's staging registers cover (20 float4 per thread at 64 channels, 40 at 128)
[[maybe_unused]] constexpr int CW_NUM_RECORDS = 0x7ffffff0;
constexpr int cw_bme(int dil) { return dil == 1 ? 256 : dil == 3 ? 252 : 240; }   // output rows per block: whole pairs, <= 128 pairs
// Tap-group kinds (see the kernel): products per channel block; accumulator set / weight slot, rows and operation of the n-th product
[[maybe_unused]] __host__ __device__ constexpr int cw_nc(int kind) { return kind == 0 ? 4 : kind == 1 ? 2 : 3; }
[[maybe_unused]] __host__ __device__ constexpr int cw_slot(int kind, int n) { return kind == 0 ? n : kind == 1 ? (n == 0 ? 0 : 3) : (n == 2 ? 3 : n); }
[[maybe_unused]] __host__ __device__ constexpr int cw_ja(int kind, int n) { return kind == 0 ? (n == 3 ? 1 : n) : kind == 1 ? n : (n == 0 ? 0 : 1); }
[[maybe_unused]] __host__ __device__ constexpr int cw_jb(int kind, int n) { return kind == 0 ? (n == 2 ? 1 : n == 3 ? 3 : 2) : kind == 1 ? n : (n == 2 ? 2 : 1); }
[[maybe_unused]] __host__ __device__ constexpr int cw_op(int kind, int n) { return kind == 0 ? (n == 1 ? 1 : -1) : kind == 1 ? 0 : (n == 1 ? 0 : -1); }   // -1: a - b, +1: a + b, 0: a
}  // namespace

// WW[co][(g * 4 + f) * C + ci] from W[co][tap * C + ci] (tap-major conv weights), taps beyond k are zero
__global__ void wino_pack_kernel(const float* __restrict__ W, float* __restrict__ WW, int C, int taps, int groups) {
  const int n = C * groups * C;                // (co, g, ci) triples
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += gridDim.x * blockDim.x) {
    const int ci = idx % C, g = (idx / C) % groups, co = idx / (C * groups);
    const float* w = W + (size_t)co * taps * C + ci;
    const float w0 = 3 * g < taps ? w[(size_t)(3 * g) * C] : 0.f;
    const float w1 = 3 * g + 1 < taps ? w[(size_t)(3 * g + 1) * C] : 0.f;
    const float w2 = 3 * g + 2 < taps ? w[(size_t)(3 * g + 2) * C] : 0.f;
    float* o = WW + (size_t)co * groups * 4 * C + (size_t)g * 4 * C + ci;
    const int left = taps - 3 * g;               // taps of this group
    if (left >= 3) {                             // F(2,3)
      o[0] = w0;
      o[C] = 0.5f * ((w0 + w1) + w2);
      o[2 * C] = 0.5f * ((w0 - w1) + w2);
      o[3 * C] = w2;
    } else if (left == 2) {                      // F(2,2): slots 0, 1, 3 (kind 2 in the kernel)
      o[0] = w0; o[C] = w0 + w1; o[2 * C] = 0.f; o[3 * C] = w1;
    } else {                                     // one tap: slots 0 and 3 (kind 1): y[t] += w0 x0, y[t + d] -= (-w0) x1
      o[0] = w0; o[C] = 0.f; o[2 * C] = 0.f; o[3 * C] = -w0;
    }
  }
}

// CH = 64: two workgroups of 4 waves per CU.  CH = 128 (the 128-channel stage, which has no direct slab kernel -- conv_sk2<128> is what this
// replaces): ONE workgroup of 8 waves per CU on one slab; waves 0-3 produce output columns 0-63, waves 4-7 columns 64-127 of the same 128
// pairs, so every wave keeps the 32-tile accumulator set, the 8-fragment ring and the register budget (256) of the 64-channel form, and
// each SIMD still holds two waves that hide each other's fragment latency (a 64-tile accumulator set per wave made hipcc shuffle
// accumulators between register classes inside the loop: 1300 moves per 2048 MFMAs, 200+ spills).
// CH = 256 (round 5: the 256-channel stage, until then on conv_sk2<128> + twins): the 128-channel form run over TWO slab phases and TWO
// column halves.  A 256-channel slab row is 1 KB -- a block with its halo does not fit the LDS -- so the contraction walks the input
// channels in two halves of 128 (slab phase kh: stage channels [128 kh, 128 kh + 128) of the block's rows, contract them into the
// SAME four accumulator sets, restage), and a workgroup produces 128 of the 256 output columns: workgroup w works column half
// (w >> 3) & 1 -- a constant of the workgroup, so its weight rows and the ring prefetch across blocks never change -- and the two
// workgroups that share a block (w and w ^ 8) sit on the same XCD (block w % 8 of the dispatch order), so the second reader of the
// block's input rows hits that XCD's L2.  Per (block, column half) twice the MFMA work and twice the staging of the 128-channel form.
template <int DIL, int CH, int TAIL>
__global__ __launch_bounds__(CH >= 128 ? 512 : 256, CH >= 128 ? 1 : CH == 64 ? 2 : 3) void conv_c64w_kernel(const GemmArgs p, const int groups, const int slab_rows) {
#if __HIP_DEVICE_COMPILE__
  // (CH = 32: the k = 11 ResBlock convs of the 32-channel stage, three workgroups per CU: 2 column tiles, 16 MFMAs per sub-step)
  constexpr int CS = CH == 256 ? 128 : CH;                                  // input channels of one slab phase
  constexpr int KH = CH / CS;                                               // slab phases
  constexpr int C = CH, LDA = CS + 4, CT = CH >= 64 ? 4 : CH / 16, CB = CS / 16, RING = 2 * CT, TPR = CS / 4;   // TPR threads stage one row
  constexpr int NT = CH >= 128 ? 512 : 256, RPP = NT / TPR;                 // rows per staging pass (16; 32 at 32 channels)
  // (CH = 256: the accumulators stay live across the second phase's staging -- 10 float4 in flight per thread, two passes)
  constexpr int NP = (CW_MAXROWS + RPP - 1) / RPP, NPC = CH == 256 ? 10 : NP < 20 ? NP : 20, BME = cw_bme(DIL), NPAIR = BME / 2;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sA = smem;                                                        // slab [slab_rows][68]
  int* s_blk = reinterpret_cast<int*>(smem + ((slab_rows * LDA + 3) & ~3));   // block prefix per segment

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane((t >> 6) & 3);
  const int cs = CH <= 64 ? 0 : __builtin_amdgcn_readfirstlane(t >> 8);     // column set of this wave
  const int colh = CH == 256 ? (int)((blockIdx.x >> 3) & 1) : 0;             // column half of this WORKGROUP (CH = 256)
  const int col0 = colh * 128 + cs * 64;                                    // first output column of this wave
  const int r = lane & 15, g = lane >> 4;
  const int Kw = groups * 4 * C;                                           // row length of the transformed weight matrix

  const int nseg = p.nseg > 0 ? p.nseg : 1;
  if (t == 0) {
    int acc = 0;
    for (int s = 0; s < nseg; ++s) {
      s_blk[s] = acc;
      const int len = p.nseg > 0 ? p.segs[4 * s + 1] : p.M;
      acc += (len + BME - 1) / BME;
    }
    s_blk[nseg] = acc;
  }
  __syncthreads();
  const int nblocks = s_blk[nseg];
  // (no input activation: slope 1 -- max(v, v * 1) = v exactly; one kernel for both, see the TAIL note below)
  const float slope = p.in_act == ACT_LRELU ? p.in_slope : 1.0f;

  int seg = 0, seg_lo = 0, seg_hi = 0, m0 = 0;
  auto locate = [&](int blk) {                 // blocks ascend per workgroup
    while (blk >= s_blk[seg + 1]) ++seg;
    seg_lo = p.nseg > 0 ? p.segs[4 * seg] : 0;
    seg_hi = seg_lo + (p.nseg > 0 ? p.segs[4 * seg + 1] : p.in_len);
    m0 = seg_lo + (blk - s_blk[seg]) * BME;    // first output row (packed coordinates)
  };

  // ---- weight fragments: L2 -> registers.  Fragment (idx = group * 4 + component, channel block cc, column tile ct) ----
  const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, CW_NUM_RECORDS, 0x00020000);
  const int vo = (r * Kw + 4 * g) * 4;
  auto wload = [&](int idx, int cc, int ct, int kh) -> f32x4 {
    const int so = __builtin_amdgcn_readfirstlane(((col0 + ct * 16) * Kw + idx * C + kh * CS + cc * 16) * 4);
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsW, vo, so, 0);
    return f32x4{__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3])};
  };

  // CH = 256: workgroups 16 a + b and 16 a + 8 + b (b < 8) share block index 8 a + b and its stride (grid % 16 == 0, host)
  int blk = CH == 256 ? (int)((blockIdx.x & 7) + 8 * (blockIdx.x >> 4)) : (int)blockIdx.x;
  const int blk_stride = CH == 256 ? (int)(gridDim.x >> 1) : (int)gridDim.x;
  if (blk >= nblocks) return;
  f32x4 ring[RING];                            // sub-steps 0 and 1 of group 0: (cc 0, f 0), (cc 0, f 1)
#pragma unroll
  for (int q = 0; q < RING; ++q) ring[q] = wload(q / CT, 0, q % CT, 0);

  // this lane's two pairs (pair tile i = 0 / 1 of the wave): first row of pair p = 2 d (p / d) + p % d, the second is + d
  int toff[2];
  bool pv[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int pp = wave * 32 + i * 16 + r;
    pv[i] = pp < NPAIR;
    const int pc = pv[i] ? pp : NPAIR - 1;
    toff[i] = 2 * DIL * (pc / DIL) + pc % DIL;
  }

#if CW_TIMING
  unsigned long long cw_acc[4] = {0, 0, 0, 0}, cw_last = __builtin_amdgcn_s_memtime(), cw_begin = cw_last, cw_blocks = 0;
#endif
  for (; blk < nblocks; blk += blk_stride) {
#if CW_TIMING
    ++cw_blocks;
#endif
    locate(blk);
    const int cm0 = m0;
    const int m_hi = p.nseg > 0 ? seg_hi : min(seg_hi, p.M);
    const bool edge = (m0 - p.pad < seg_lo) || (m0 - p.pad + slab_rows > seg_hi);
    f32x4 acc[4][2][CT];                                   // [component][pair tile][column tile]
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < CT; ++j) acc[f][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int kh = 0; kh < KH; ++kh) {                       // slab phases (one, except CH = 256: two halves of the input channels)
    const int kh_wrap = kh + 1 < KH ? kh + 1 : 0;          // phase of the fragments requested past this phase's last group
    __syncthreads();                                       // previous slab's reads are done
    CW_STAMP(kh == 0 ? 3 : 1);                             // (phase 0: the wait for the other waves' epilogue counts as wait)
    // (Requesting the first pass's loads BEFORE this barrier -- so that a wave that finished early overlaps the load latency with its
    //  wait -- was built and measured: the time only moves from "staging" to "wait", the loads are bound by the burst of all CUs staging
    //  at once, not by when they are issued: profiles/r05_cw_early_load_experiment.txt, bench -0.5 %.)
    // ---- slab: global -> registers -> [zero padding, leaky-ReLU] -> LDS (as conv_c64.hip), 20 float4 per thread in flight at a time ----
#pragma unroll 1
    for (int u0 = 0; u0 < NP; u0 += NPC) {
      f32x4 pre[NPC];
#pragma unroll
      for (int u = 0; u < NPC; ++u) {
        const int rho = t / TPR + RPP * (u0 + u);
        const int gc = min(max(m0 - p.pad + rho, seg_lo), seg_hi - 1);
        pre[u] = *reinterpret_cast<const f32x4*>(p.A + (size_t)gc * p.lda + kh * CS + (t % TPR) * 4);
      }
      float* dst = sA + (t / TPR + RPP * u0) * LDA + (t % TPR) * 4;
#pragma unroll
      for (int u = 0; u < NPC; ++u) {
        const int rho = t / TPR + RPP * (u0 + u);
        f32x4 v = pre[u];
        if (edge) {
          const int gin = m0 - p.pad + rho;
          const bool ok = gin >= seg_lo && gin < seg_hi;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = ok ? v[e] : 0.f;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], v[e] * slope);
        if (rho < slab_rows) *reinterpret_cast<f32x4*>(dst + u * RPP * LDA) = v;
      }
    }
    __syncthreads();
    CW_STAMP(0);

    // A tap group is one of three KINDS (rows in units of d from the group's first input row x0; products accumulate in the four
    // accumulator sets m0..m3, the output transform is y[t] = m0 + m1 + m2, y[t + d] = m1 - m2 - m3 for every kind):
    //   kind 0, three taps (F(2,3)): m0 += G0 (x0 - x2), m1 += G1 (x1 + x2), m2 += G2 (x2 - x1), m3 += G3 (x1 - x3)          4 products
    //   kind 1, ONE tap left (k = 7): y[t] += w0 x0, y[t + d] += w0 x1   =>   m0 += w0 . x0,  m3 += (-w0) . x1                  2 products
    //   kind 2, TWO taps left (k = 11): F(2,2):  m0 += w0 (x0 - x1),  m1 += (w0 + w1) x1,  m3 += w1 (x1 - x2)                  3 products
    // (round 4 ran the incomplete last group as a zero-padded F(2,3): 4 products, one of them against an all-zero matrix at k = 11, two
    //  redundant at k = 7 -- 12 -> 10 MFMA k-blocks per pair at k = 7, 16 -> 15 at k = 11; wino_pack_kernel writes the matching weights.)
    // (tables as constexpr functions of (kind, n): cw_nc / cw_slot / cw_ja / cw_jb / cw_op above -- local arrays captured by the lambda
    //  were indexed at run time, which put the accumulators into scratch)
    const float* pa0 = sA + toff[0] * LDA + 4 * g;         // + group * 3 d rows + j d rows + cc * 16
    const float* pa1 = sA + toff[1] * LDA + 4 * g;
    auto rd = [&](const float* base, int j, int cc) -> f32x4 { return *reinterpret_cast<const f32x4*>(base + j * DIL * LDA + cc * 16); };
    auto xform = [&](int op, const f32x4 a, const f32x4 b) -> f32x4 { return op == 1 ? a + b : op == 0 ? a : a - b; };
    f32x4 xa[2];                                           // (the first group of a phase is always a full one: taps >= 3)
    xa[0] = xform(-1, rd(pa0, 0, 0), rd(pa0, 2, 0));
    xa[1] = xform(-1, rd(pa1, 0, 0), rd(pa1, 2, 0));
    // One tap group, fully unrolled over its sub-steps (channel block cc, product n).  kind_next: kind of the group that follows (the
    // sub-step after this group's last one is that group's (cc 0, product 0); the fragments requested two sub-steps ahead are its
    // products 0 and 1); pn0 / pn1, grp_next, kh_next: where that group's rows and weights are.
    auto group = [&](auto KIND, const int grp, const int kind_next, const float* pn0, const float* pn1, const int grp_next, const int kh_next) __attribute__((always_inline)) {
      constexpr int K = decltype(KIND)::value, NC = cw_nc(K), NS = CB * NC;
      static_assert(NS % 2 == 0 && NS >= 2, "ring slots keep their parity across groups");
#pragma unroll
      for (int ss = 0; ss < NS; ++ss) {
        const int slot = cw_slot(K, ss % NC);
        f32x4 na0, nb0, na1, nb1;
        if (ss + 1 < NS) {
          const int ccn = (ss + 1) / NC, nn = (ss + 1) % NC;
          na0 = rd(pa0, cw_ja(K, nn), ccn); nb0 = rd(pa0, cw_jb(K, nn), ccn);
          na1 = rd(pa1, cw_ja(K, nn), ccn); nb1 = rd(pa1, cw_jb(K, nn), ccn);
        } else {                                           // product 0 of the next group: x0 - x2 | x0 | x0 - x1
          const int jb = kind_next == 0 ? 2 : 1;
          na0 = rd(pn0, 0, 0); nb0 = *reinterpret_cast<const f32x4*>(pn0 + jb * (DIL * LDA));
          na1 = rd(pn1, 0, 0); nb1 = *reinterpret_cast<const f32x4*>(pn1 + jb * (DIL * LDA));
        }
        f32x4 wf[CT];
#pragma unroll
        for (int j = 0; j < CT; ++j) {
          const int q = ss * CT + j;                       // fragment of this sub-step; its ring slot is re-armed two sub-steps ahead
          wf[j] = ring[q % RING];
          const int s2 = ss + 2;
          if (s2 < NS) ring[q % RING] = wload(grp * 4 + cw_slot(K, s2 % NC), s2 / NC, j, kh);
          else ring[q % RING] = wload(grp_next * 4 + (s2 == NS ? 0 : (kind_next == 1 ? 3 : 1)), 0, j, kh_next);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < CT; ++j)
              acc[slot][i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[j][e], xa[i][e], acc[slot][i][j], 0, 0, 0);   // D = G . D^T
        if (ss + 1 < NS) {
          xa[0] = xform(cw_op(K, (ss + 1) % NC), na0, nb0);
          xa[1] = xform(cw_op(K, (ss + 1) % NC), na1, nb1);
        } else {
          xa[0] = kind_next == 1 ? na0 : na0 - nb0;
          xa[1] = kind_next == 1 ? na1 : na1 - nb1;
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    // TAIL (taps % 3) is a template argument: with both tail bodies behind run-time branches in one kernel hipcc spilled 280
    // registers (each alone: none) -- the template argument LRELU of round 4 made room for it (36 instantiations instead of 72)
    constexpr int tail = TAIL;
    const int groups_full = groups - (tail ? 1 : 0);
#pragma unroll 1
    for (int grp = 0; grp < groups_full; ++grp) {
      const bool more = grp + 1 < groups;                              // another group of this phase follows
      const float* pn0 = more ? pa0 + 3 * DIL * LDA : pa0;             // (after the last group: a harmless re-read)
      const float* pn1 = more ? pa1 + 3 * DIL * LDA : pa1;
      // after the phase's last group: the next phase's / the next block's first fragments (a full group)
      group(std::integral_constant<int, 0>{}, grp, (more && grp + 1 == groups_full) ? tail : 0, pn0, pn1, more ? grp + 1 : 0, more ? kh : kh_wrap);
      pa0 = pn0;
      pa1 = pn1;
    }
    if constexpr (TAIL == 1) group(std::integral_constant<int, 1>{}, groups_full, 0, pa0, pa1, 0, kh_wrap);
    if constexpr (TAIL == 2) group(std::integral_constant<int, 2>{}, groups_full, 0, pa0, pa1, 0, kh_wrap);
    CW_STAMP(1);
    }   // slab phases

    // ---- output transform + epilogue: lane holds 4 consecutive channels (4g .. 4g+3 of column tile j) of both rows of pair r ----
    int le = lane;
    asm volatile("" : "+v"(le));               // addresses derived from `le` cannot be hoisted above the contraction
    const int g_e = le >> 4;
    f32x4 bb[CT];
#pragma unroll
    for (int j = 0; j < CT; ++j) {
      bb[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (p.bias) bb[j] = *reinterpret_cast<const f32x4*>(p.bias + col0 + j * 16 + g_e * 4);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {                        // the pair's first / second row
        const int m = cm0 + toff[i] + h * DIL;
        const int mc = min(m, m_hi - 1);
        f32x4 rr[CT], rr2[CT];
        if (p.R) {
#pragma unroll
          for (int j = 0; j < CT; ++j) rr[j] = *reinterpret_cast<const f32x4*>(p.R + (size_t)mc * p.ldr + col0 + j * 16 + g_e * 4);
        }
        if (p.R2) {
#pragma unroll
          for (int j = 0; j < CT; ++j) rr2[j] = *reinterpret_cast<const f32x4*>(p.R2 + (size_t)mc * p.ldr2 + col0 + j * 16 + g_e * 4);
        }
#pragma unroll
        for (int j = 0; j < CT; ++j) {
          const int n = col0 + j * 16 + g_e * 4;
          f32x4 v = h == 0 ? (acc[0][i][j] + acc[1][i][j]) + acc[2][i][j] : (acc[1][i][j] - acc[2][i][j]) - acc[3][i][j];
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
          if (pv[i] && m < m_hi) {
            *reinterpret_cast<f32x4*>(p.C + (size_t)m * p.ldc + n) = v;
            if (CH >= 128 && p.C2) {                         // pre-activated twin for a consumer that cannot activate while staging (conv_sk2)
              f32x4 w2;
#pragma unroll
              for (int e = 0; e < 4; ++e) w2[e] = v[e] > 0.f ? v[e] : v[e] * p.c2_slope;
              *reinterpret_cast<f32x4*>(p.C2 + (size_t)m * p.ldc2 + n) = w2;
            }
          }
        }
      }
    }
    CW_STAMP(2);
  }
#if CW_TIMING
  if (t == 0) {
    unsigned long long* d = g_cw_dbg + (size_t)(blockIdx.x & 1023) * 8;
    d[0] = cw_acc[0]; d[1] = cw_acc[1]; d[2] = cw_acc[2]; d[3] = cw_acc[3]; d[4] = cw_blocks; d[5] = cw_begin; d[6] = __builtin_amdgcn_s_memtime(); d[7] = 1;
  }
#endif
#endif
}

#if CW_TIMING
// [wg][8] = staging, contraction, epilogue, wait-at-block-start cycles, blocks, t_begin, t_end, valid
extern "C" int ss_debug_cw_timing(unsigned long long* h_out, int cap_wgs) {
  const int n = cap_wgs < 1024 ? cap_wgs : 1024;
  if (hipMemcpyFromSymbol(h_out, HIP_SYMBOL(g_cw_dbg), (size_t)n * 8 * sizeof(unsigned long long)) != hipSuccess) return -1;
  static unsigned long long zeros[1024 * 8];
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_cw_dbg), zeros, sizeof(zeros));
  return n;
}
#endif

// ---- host side ---------------------------------------------------------------------------------
// (A/B knobs SS_CONV_C64 / C128 / C256 / C32_WINOGRAD, ..._MIN_K, ..._MIN_ROWS and these hooks: dispatch.hpp)
void conv_c64w_debug(int enable) { if (enable >= 0) dispatch_edit([enable](Dispatch& d) { d.c64w_on = enable ? 1 : 0; }); }
bool conv_c64w_enabled() { return disp().c64w_on != 0; }
void conv_c128w_debug(int enable) { if (enable >= 0) dispatch_edit([enable](Dispatch& d) { d.c128w_on = enable ? 1 : 0; }); }
bool conv_c128w_enabled() { return disp().c128w_on != 0; }

void conv_c256w_debug(int enable) { if (enable >= 0) dispatch_edit([enable](Dispatch& d) { d.c256w_on = enable ? 1 : 0; }); }
bool conv_c256w_enabled() { return disp().c256w_on != 0; }

static int cw_groups(const GemmArgs& a) { return (a.taps + 2) / 3; }
static size_t cw_lds(const GemmArgs& a, int ch) {
  const int slab_rows = cw_bme(a.dil) + 3 * cw_groups(a) * a.dil;
  return (size_t)((slab_rows * (ch + 4) + 3) & ~3) * sizeof(float) + (CW_MAXSEG + 2) * sizeof(int);
}

// the launches conv_c64.hip takes (checked by the caller: conv_c64_eligible) that also have transformed weights, a "same" geometry the
// pairing covers and a slab two of which fit a CU
bool conv_c64w_eligible(const GemmArgs& a) {
  if (!disp().c64w_on || !a.Wwino || a.taps < disp().c64w_min_k || a.taps < 3 || (a.dil != 1 && a.dil != 3 && a.dil != 5) || a.C2) return false;
  if (a.pad != a.dil * (a.taps - 1) / 2) return false;
  // k = 7 at dilation 5: 1.17x fewer MFMAs against a 45-row halo on 240-row blocks -- measured slower than the direct form (283 vs 272 us)
  static const int allow_k7d5 = getenv("SS_CONV_C64_WINOGRAD_K7D5") ? atoi(getenv("SS_CONV_C64_WINOGRAD_K7D5")) : 0;
  if (a.taps <= 8 && a.taps > 3 && a.dil == 5 && !allow_k7d5) return false;
  const int slab_rows = cw_bme(a.dil) + 3 * cw_groups(a) * a.dil;
  return slab_rows <= CW_MAXROWS && 2 * cw_lds(a, 64) <= 158 * 1024 && (size_t)cw_groups(a) * 4 * 64 * 64 * 4 < 0x7ff00000ull;
}

// The 128-channel stage: every "same" conv with C = N = 128, k >= 3 at dilation 1 / 3 / 5 of a packed batch big enough to give each CU a
// block (one workgroup per CU: the slab is 137-158 KB).  There is no direct slab kernel at this width (measured slower than conv_sk2<128>:
// profiles/r04_c128_bench.txt), so the stage takes this path only if ALL its convs are eligible (model.hip asks with a probe).
bool conv_c128w_eligible(const GemmArgs& a) {
  if (!disp().c128w_on || !a.Wwino || !a.same_rows || a.stride != 1 || a.chunk || a.glu || a.ln_g || a.x3 || a.Cin != 128 || a.N != 128) return false;
  if (a.lda != 128 || (a.ldc & 3) || (a.R && (a.ldr & 3)) || (a.R2 && (a.ldr2 & 3)) || (a.C2 && (a.ldc2 & 3))) return false;
  if (a.taps < 3 || (a.dil != 1 && a.dil != 3 && a.dil != 5) || a.pad != a.dil * (a.taps - 1) / 2) return false;
  if (a.nseg > CW_MAXSEG || a.M < disp().c128w_min_rows || !slab_rows_ok(a.M)) return false;
  if (!(a.in_act == ACT_NONE || (a.in_act == ACT_LRELU && a.in_slope > 0.f && a.in_slope < 1.f)) || !(a.act == ACT_NONE || a.act == ACT_LRELU)) return false;
  const int slab_rows = cw_bme(a.dil) + 3 * cw_groups(a) * a.dil;
  return slab_rows <= CW_MAXROWS && cw_lds(a, 128) <= 160 * 1024;
}

// The 256-channel stage in the same form (two slab phases of 128 input channels, two column halves): all-or-nothing per stage like the
// 128-channel one (model.hip probes every conv).
bool conv_c256w_eligible(const GemmArgs& a) {
  if (!disp().c256w_on || !a.Wwino || !a.same_rows || a.stride != 1 || a.chunk || a.glu || a.ln_g || a.x3 || a.Cin != 256 || a.N != 256) return false;
  if (a.lda != 256 || (a.ldc & 3) || (a.R && (a.ldr & 3)) || (a.R2 && (a.ldr2 & 3)) || (a.C2 && (a.ldc2 & 3))) return false;
  if (a.taps < 3 || (a.dil != 1 && a.dil != 3 && a.dil != 5) || a.pad != a.dil * (a.taps - 1) / 2) return false;
  if (a.nseg > CW_MAXSEG || a.M < disp().c256w_min_rows || !slab_rows_ok(a.M)) return false;
  if ((size_t)256 * cw_groups(a) * 4 * 256 * 4 >= 0x7ff00000ull) return false;
  if (!(a.in_act == ACT_NONE || (a.in_act == ACT_LRELU && a.in_slope > 0.f && a.in_slope < 1.f)) || !(a.act == ACT_NONE || a.act == ACT_LRELU)) return false;
  const int slab_rows = cw_bme(a.dil) + 3 * cw_groups(a) * a.dil;
  return slab_rows <= CW_MAXROWS && cw_lds(a, 128) <= 160 * 1024;
}

int launch_wino_pack(const float* W, float* WW, int C, int taps, hipStream_t stream) {
  const int groups = (taps + 2) / 3, n = C * groups * C;
  hipLaunchKernelGGL(wino_pack_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, W, WW, C, taps, groups);
  SS_LAUNCH_CHECK();
  return SS_OK;
}

template <int DIL, int CH, int TAIL>
static int launch_cw_t(GemmArgs a, hipStream_t stream) {
  constexpr int BME = cw_bme(DIL);
  const int groups = cw_groups(a);
  const int slab_rows = BME + 3 * groups * DIL;
  const size_t lds = cw_lds(a, CH == 256 ? 128 : CH);
  SS_MAX_LDS_ONCE((&conv_c64w_kernel<DIL, CH, TAIL>), CH >= 128 ? 160 * 1024 : 96 * 1024);
  SkWorkspace* st = nullptr;                       // (only for the device's CU count, cached per context)
  int rc = sk_workspace_acquire(stream, &st);
  if (rc != SS_OK) return rc;
  const int nseg = a.nseg > 0 ? a.nseg : 1;
  const long long max_blocks = (long long)cdiv(a.M, BME) + nseg;      // upper bound (per-segment round-up)
  int grid = (int)std::min<long long>((CH >= 128 ? 1ll : CH == 64 ? 2ll : 3ll) * st->cus, std::max<long long>(1, max_blocks));
  if (CH == 256) {                                 // (block, column half) items: workgroups come in groups of 16 = 8 blocks x 2 halves
    const long long want = std::min<long long>(st->cus, 2 * ((max_blocks + 7) / 8 * 8));
    grid = (int)std::max<long long>(16, want / 16 * 16);
  }
  ProfRec rec{}; bool prof = false;
  rc = prof_begin(a, stream, CH == 64 ? 27 : CH == 128 ? 28 : CH == 256 ? 30 : 29, rec, prof);   // census: the conv's algorithmic (direct-form) FLOPs; the kernel issues 4 G / (2 k) of them
  if (rc != SS_OK) return rc;
  a.W = a.Wwino;
  hipLaunchKernelGGL((conv_c64w_kernel<DIL, CH, TAIL>), dim3(grid), dim3(CH >= 128 ? 512 : 256), lds, stream, a, groups, slab_rows);
  SS_LAUNCH_CHECK();
  return prof_end(stream, rec, prof);
}

template <int CH>
static int launch_cw(const GemmArgs& a, hipStream_t stream) {
  if (a.in_act != ACT_NONE && a.in_act != ACT_LRELU) return SS_ERR_ARG;
  const int key = a.dil * 10 + a.taps % 3;
  switch (key) {
    case 10: return launch_cw_t<1, CH, 0>(a, stream);
    case 11: return launch_cw_t<1, CH, 1>(a, stream);
    case 12: return launch_cw_t<1, CH, 2>(a, stream);
    case 30: return launch_cw_t<3, CH, 0>(a, stream);
    case 31: return launch_cw_t<3, CH, 1>(a, stream);
    case 32: return launch_cw_t<3, CH, 2>(a, stream);
    case 50: return launch_cw_t<5, CH, 0>(a, stream);
    case 51: return launch_cw_t<5, CH, 1>(a, stream);
    case 52: return launch_cw_t<5, CH, 2>(a, stream);
    default: return SS_ERR_ARG;
  }
}

int launch_conv_c64w(const GemmArgs& a, hipStream_t stream) {
  if (!conv_c64_eligible(a) || !conv_c64w_eligible(a)) return SS_ERR_ARG;
  return launch_cw<64>(a, stream);
}
// The 32-channel stage's per-conv launches (model.hip: k >= 11) -- the convs conv_c32.hip takes, with transformed weights
void conv_c32w_debug(int enable) { if (enable >= 0) dispatch_edit([enable](Dispatch& d) { d.c32w_on = enable ? 1 : 0; }); }
bool conv_c32w_enabled() { return disp().c32w_on != 0; }
bool conv_c32w_eligible(const GemmArgs& a) {                 // call with conv_c32_eligible(a) already true
  if (!disp().c32w_on || !a.Wwino || a.taps < 3 || (a.dil != 1 && a.dil != 3 && a.dil != 5) || a.C2) return false;
  if (a.pad != a.dil * (a.taps - 1) / 2) return false;
  const int slab_rows = cw_bme(a.dil) + 3 * cw_groups(a) * a.dil;
  return slab_rows <= CW_MAXROWS && 3 * cw_lds(a, 32) <= 158 * 1024;
}
int launch_conv_c32w(const GemmArgs& a, hipStream_t stream) {
  if (!conv_c32_eligible(a) || !conv_c32w_eligible(a)) return SS_ERR_ARG;
  return launch_cw<32>(a, stream);
}
int launch_conv_c128w(const GemmArgs& a, hipStream_t stream) {
  if (!conv_c128w_eligible(a)) return SS_ERR_ARG;
  return launch_cw<128>(a, stream);
}
int launch_conv_c256w(const GemmArgs& a, hipStream_t stream) {
  if (!conv_c256w_eligible(a)) return SS_ERR_ARG;
  return launch_cw<256>(a, stream);
}

}  // namespace ss
