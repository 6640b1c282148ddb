// Attention, wavefront-softmax form (see attention.hpp).
//
// Workgroup = 4 wave64 = 16 query rows of one head; wave w owns rows 4w..4w+3.  Keys are walked
// in tiles of 64 (lane = key).  K / V (and, for the rel-pos form, the 79 positional rows the
// 16x64 (query,key) pairs of the tile can touch) are staged in LDS; the query rows live in
// registers (lane d holds q[.,d]) and are broadcast with v_readlane, as are the probabilities
// for P.V (lane = output dim).  Softmax is the online (running max / running sum) form with
// wave-wide shuffles, so the [T,T] and [T,2T-1] score matrices the reference materialises
// (espnet_multihead_attention.py:186-196) never exist.
#include "attention.hpp"

#include <algorithm>
#include <cstdlib>

namespace ss {

void attention_debug_no_mfma(int v) { dispatch_edit([v](Dispatch& d) { d.attn_no_mfma = v; }); }   // test hook: route plain attention to the VALU kernel
void attention_debug_q16(int v) { dispatch_edit([v](Dispatch& d) { d.attn_q16 = v; }); }          // test hook: 0 = the few-queries form off
void attention_debug_split(int v) { dispatch_edit([v](Dispatch& d) { d.attn_split = v; }); }   // test hook; SS_ATTN_NO_SPLIT=1: never split

constexpr int QB = 16;     // query rows per workgroup
constexpr int KT = 64;     // keys per tile
constexpr int DH = 64;     // head dim
constexpr int LDKS = DH + 1;

__device__ __forceinline__ float rdlane(float v, int l) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}

template <bool RELPOS>
__global__ __launch_bounds__(256) void attention_kernel(AttnArgs p) {
  if (p.nseg > 0) {   // ragged batch: rebase this workgroup onto its utterance
    const int* sg = p.segs + 4 * blockIdx.z;
    p.Tq = sg[1]; p.Tk = sg[3];
    if ((int)blockIdx.x * QB >= p.Tq) return;
    p.Q += (size_t)sg[0] * p.ldq; p.O += (size_t)sg[0] * p.ldo;
    p.K += (size_t)sg[2] * p.ldk; p.V += (size_t)sg[2] * p.ldv;
    if (RELPOS) p.P += (size_t)(p.p_tmax - p.Tk) * p.ldp;
  }
  __shared__ float Ks[KT * LDKS];
  __shared__ float Vs[KT * DH];
  __shared__ float Ps[RELPOS ? (KT + QB - 1) * LDKS : 1];

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int h = blockIdx.y;
  const int i0 = blockIdx.x * QB;
  const int hoff = h * DH;
  const int qoff = p.Tk - p.Tq;
  const int q0 = p.q0;               // absolute position of query row 0 (chunk mask / relative offsets)

  // query rows of this wave in registers: lane d holds element d
  float qu[4], qv[4];
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    const int i = i0 + wave * 4 + rr;
    float q = (i < p.Tq) ? p.Q[(size_t)i * p.ldq + hoff + lane] : 0.f;
    if (RELPOS) {
      qu[rr] = q + p.bias_u[hoff + lane];
      qv[rr] = q + p.bias_v[hoff + lane];
    } else {
      qu[rr] = q;
      qv[rr] = 0.f;
    }
  }

  float m_run[4], l_run[4], acc[4];
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) { m_run[rr] = -INFINITY; l_run[rr] = 0.f; acc[rr] = 0.f; }

  // last key any row of this block may see (block-uniform loop bound)
  int kmax = p.Tk - p.k_mask_tail;
  const int ilast = min(i0 + QB, p.Tq) - 1;
  if (p.causal) kmax = min(kmax, ilast + qoff + 1);
  if (p.chunk > 0) kmax = min(kmax, ((ilast + q0) / p.chunk + 1) * p.chunk);

  for (int j0 = 0; j0 < kmax; j0 += KT) {
    __syncthreads();  // previous tile fully consumed
    // stage K, V tiles: 64 rows x 16 float4
    for (int f = t; f < KT * (DH / 4); f += 256) {
      const int row = f >> 4, c4 = (f & 15) * 4;
      const int j = j0 + row;
      float4 kv = make_float4(0.f, 0.f, 0.f, 0.f), vv = kv;
      if (j < p.Tk) {
        kv = *reinterpret_cast<const float4*>(p.K + (size_t)j * p.ldk + hoff + c4);
        vv = *reinterpret_cast<const float4*>(p.V + (size_t)j * p.ldv + hoff + c4);
      }
      float* kd = Ks + row * LDKS + c4;
      kd[0] = kv.x; kd[1] = kv.y; kd[2] = kv.z; kd[3] = kv.w;
      *reinterpret_cast<float4*>(Vs + row * DH + c4) = vv;
    }
    if (RELPOS) {
      // local row lr <-> table row pbase + lr, pbase = j0 - (i0 + QB - 1) + Tk - 1
      const int pbase = j0 - (i0 + q0 + QB - 1) + p.Tk - 1;
      for (int f = t; f < (KT + QB - 1) * (DH / 4); f += 256) {
        const int row = f >> 4, c4 = (f & 15) * 4;
        const int pr = pbase + row;
        float4 pv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (pr >= 0 && pr < 2 * p.Tk - 1)
          pv = *reinterpret_cast<const float4*>(p.P + (size_t)pr * p.ldp + hoff + c4);
        float* pd = Ps + row * LDKS + c4;
        pd[0] = pv.x; pd[1] = pv.y; pd[2] = pv.z; pd[3] = pv.w;
      }
    }
    __syncthreads();

    // scores: lane = key
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    const float* krow = Ks + lane * LDKS;
#pragma unroll 16
    for (int d = 0; d < DH; ++d) {
      const float kd = krow[d];
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) s[rr] = fmaf(rdlane(qu[rr], d), kd, s[rr]);
    }
    if (RELPOS) {
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        // table row for (i, j): j - i + Tk - 1  ->  local row lane + (QB-1) - (wave*4+rr)
        const float* prow = Ps + (lane + (QB - 1) - (wave * 4 + rr)) * LDKS;
        float b = 0.f;
#pragma unroll 16
        for (int d = 0; d < DH; ++d) b = fmaf(rdlane(qv[rr], d), prow[d], b);
        s[rr] += b;
      }
    }

    const int j = j0 + lane;
    float pr[4];
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int i = i0 + wave * 4 + rr;
      bool vis = (j < p.Tk - p.k_mask_tail) && (i < p.Tq);
      if (p.causal) vis = vis && (j <= i + qoff);
      if (p.chunk > 0) vis = vis && (j < ((i + q0) / p.chunk + 1) * p.chunk);
      const float sv = vis ? s[rr] * p.scale : -INFINITY;
      const float mt = wave_max(sv);
      const float mn = fmaxf(m_run[rr], mt);
      float pe = 0.f, corr = 1.f;
      if (mn > -INFINITY) {
        pe = vis ? expf(sv - mn) : 0.f;
        corr = (m_run[rr] > -INFINITY) ? expf(m_run[rr] - mn) : 0.f;
      }
      l_run[rr] = l_run[rr] * corr + wave_sum(pe);
      acc[rr] *= corr;
      m_run[rr] = mn;
      pr[rr] = pe;
    }
    // P.V : lane = output dim
#pragma unroll 16
    for (int jj = 0; jj < KT; ++jj) {
      const float vd = Vs[jj * DH + lane];
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) acc[rr] = fmaf(rdlane(pr[rr], jj), vd, acc[rr]);
    }
  }

#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    const int i = i0 + wave * 4 + rr;
    if (i < p.Tq) p.O[(size_t)i * p.ldo + hoff + lane] = acc[rr] / l_run[rr];
  }
}


// -------------------------------------------------------------------------------------------------
// Decode form (Tq <= 8, e.g. one new MT token against the KV cache): one wave64 per (query, head),
// no LDS, no barriers.  Scores: lane = key, the 64-d query is wave-uniform (scalar loads), each lane
// streams its key row (256 contiguous bytes).  P.V: lane = output dim, probabilities broadcast with
// v_readlane, V rows read coalesced.  Same online softmax as the tiled kernel.
// -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attention_decode_kernel(AttnArgs p) {
  // 4 waves per (query, head): wave w takes key tiles w, w+4, ...; partial (max, sum, acc) merged in LDS
  if (p.nseg > 0) {
    const int* sg = p.segs + 4 * blockIdx.z;
    p.Tq = sg[1]; p.Tk = sg[3];
    if ((int)blockIdx.x >= p.Tq) return;
    p.Q += (size_t)sg[0] * p.ldq; p.O += (size_t)sg[0] * p.ldo;
    p.K += (size_t)sg[2] * p.ldk; p.V += (size_t)sg[2] * p.ldv;
  }
  __shared__ float part_m[4], part_l[4], part_acc[4][DH];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = blockIdx.x, h = blockIdx.y;
  const int hoff = h * DH;
  const int qoff = p.Tk - p.Tq;
  const float* q = p.Q + (size_t)i * p.ldq + hoff;
  int kmax = p.Tk - p.k_mask_tail;
  if (p.causal) kmax = min(kmax, i + qoff + 1);
  if (p.chunk > 0) kmax = min(kmax, (i / p.chunk + 1) * p.chunk);
  float m_run = -INFINITY, l_run = 0.f, acc = 0.f;
  for (int j0 = wave * 64; j0 < kmax; j0 += 256) {
    const int j = j0 + lane;
    const bool vis = j < kmax;
    // The tile's V rows are requested BEFORE its scores are formed (their addresses do not depend on the scores), so the
    // kernel pays one memory latency per tile instead of two -- it is launched ~8 times per MT decode step, each a link
    // of a dependent chain.  Rows past the last visible key are clamped to it: their probability is exactly 0, and
    // fmaf(0, v, acc) == acc, so the sum is the one the 16-row groups of the earlier form produced.
    float vv[64];
#pragma unroll
    for (int u = 0; u < 64; ++u) vv[u] = p.V[(size_t)min(j0 + u, kmax - 1) * p.ldv + hoff + lane];
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (vis) {
      const float4* kr = reinterpret_cast<const float4*>(p.K + (size_t)j * p.ldk + hoff);
#pragma unroll
      for (int d4 = 0; d4 < DH / 4; ++d4) {
        const float4 kv = kr[d4];
        s0 = fmaf(q[4 * d4 + 0], kv.x, s0); s1 = fmaf(q[4 * d4 + 1], kv.y, s1);
        s2 = fmaf(q[4 * d4 + 2], kv.z, s2); s3 = fmaf(q[4 * d4 + 3], kv.w, s3);
      }
    }
    const float sv = vis ? ((s0 + s1) + (s2 + s3)) * p.scale : -INFINITY;
    const float mn = fmaxf(m_run, wave_max(sv));
    const float pe = vis ? expf(sv - mn) : 0.f;
    const float corr = (m_run > -INFINITY) ? expf(m_run - mn) : 0.f;
    l_run = l_run * corr + wave_sum(pe);
    acc *= corr;
    m_run = mn;
#pragma unroll
    for (int u = 0; u < 64; ++u) acc = fmaf(rdlane(pe, u), vv[u], acc);
  }
  if (lane == 0) { part_m[wave] = m_run; part_l[wave] = l_run; }
  part_acc[wave][lane] = acc;
  __syncthreads();
  if (wave == 0) {
    float mt = fmaxf(fmaxf(part_m[0], part_m[1]), fmaxf(part_m[2], part_m[3]));
    float l = 0.f, o = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float f = (part_m[w] > -INFINITY) ? expf(part_m[w] - mt) : 0.f;
      l += part_l[w] * f;
      o += part_acc[w][lane] * f;
    }
    p.O[(size_t)i * p.ldo + hoff + lane] = o / l;
  }
}

// -------------------------------------------------------------------------------------------------
// MFMA form (plain attention, Tq > 8): QK^T and PV on v_mfma_f32_16x16x4_f32, online softmax on the
// accumulator fragments.  Workgroup = 64 queries x 1 head (4 waves x 16 queries), 64-key tiles of K
// and V^T in LDS shared by the 4 waves.
//
// Fragment algebra (C/D layout: col = lane&15, row = 4*(lane>>4) + reg; lane = (r, g)):
//   S^T = K . Q^T : A = K rows (lane (r,g) reads K[key r][16kk+4g .. +3] with one ds_read_b128 --
//     MFMA e then contracts d = {16kk+4g'+e}, a permutation of d shared with the Q operand),
//     B = Q rows held in registers.  The tile comes out as lane (r = query, g) holding the scores of
//     keys 4g..4g+3: a query's row lives in the 4 lanes {r, r+16, r+32, r+48} -> row max / sum are
//     2 shuffles, no LDS.
//   O^T = V^T . P^T : B = P^T is EXACTLY what the lane already holds (MFMA e takes keys {4g'+e}),
//     A = V^T[d = r][keys 4g..4g+3] is one ds_read_b128 from the transposed V tile.  The output
//     comes out as lane (r = query, g) holding d = 16dt+4g..+3 -> float4 stores.
// Same arithmetic as the reference fairseq MHA (ctc_unity/modules/multihead_attention.py:544-760):
// fp32 scores, -inf masks (causal / key padding), softmax, PV; only the summation order differs.
// -------------------------------------------------------------------------------------------------
constexpr int MQ = 64;              // queries per workgroup
constexpr int LDT = 68;             // padded LDS row (floats): 17 sixteen-byte slots -> conflict-free b128 fragments

__global__ __launch_bounds__(256) void attention_mfma_kernel(AttnArgs p) {
  if (p.nseg > 0) {
    const int* sg = p.segs + 4 * blockIdx.z;
    p.Tq = sg[1]; p.Tk = sg[3];
    if ((int)blockIdx.x * MQ >= p.Tq) return;
    p.Q += (size_t)sg[0] * p.ldq; p.O += (size_t)sg[0] * p.ldo;
    p.K += (size_t)sg[2] * p.ldk; p.V += (size_t)sg[2] * p.ldv;
  }
  __shared__ __attribute__((aligned(16))) float Ks[KT * LDT];     // [key][d]
  __shared__ __attribute__((aligned(16))) float Vt[DH * LDT];     // [d][key]
  using f32x4 = __attribute__((ext_vector_type(4))) float;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int h = blockIdx.y, hoff = h * DH;
  const int i0 = blockIdx.x * MQ;
  const int qoff = p.Tk - p.Tq;
  const int iq = i0 + wave * 16 + r;                 // this lane's query row
  const bool q_ok = iq < p.Tq;

  f32x4 qf[4];                                        // Q[iq][16kk + 4g .. +3]
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    qf[kk] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (q_ok) qf[kk] = *reinterpret_cast<const f32x4*>(p.Q + (size_t)iq * p.ldq + hoff + 16 * kk + 4 * g);
  }
  f32x4 o[4];                                         // O^T fragments: d = 16dt + 4g + e for query r
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY, l_run = 0.f;

  int kmax = p.Tk - p.k_mask_tail;                    // block-uniform loop bound
  const int ilast = min(i0 + MQ, p.Tq) - 1;
  if (p.causal) kmax = min(kmax, ilast + qoff + 1);
  if (p.chunk > 0) kmax = min(kmax, (ilast / p.chunk + 1) * p.chunk);
  int lim = p.Tk - p.k_mask_tail;                     // this lane's own visibility bound
  if (p.causal) lim = min(lim, iq + qoff + 1);
  if (p.chunk > 0) lim = min(lim, (iq / p.chunk + 1) * p.chunk);
  if (!q_ok) lim = 0;

  for (int j0 = 0; j0 < kmax; j0 += KT) {
    __syncthreads();                                  // previous tile fully consumed
    for (int f = t; f < KT * (DH / 4); f += 256) {
      const int row = f >> 4, c4 = (f & 15) * 4;      // 16 threads per key row (256 contiguous bytes)
      const int j = j0 + row;
      f32x4 kv = {0.f, 0.f, 0.f, 0.f}, vv = kv;
      if (j < p.Tk) {
        kv = *reinterpret_cast<const f32x4*>(p.K + (size_t)j * p.ldk + hoff + c4);
        vv = *reinterpret_cast<const f32x4*>(p.V + (size_t)j * p.ldv + hoff + c4);
      }
      *reinterpret_cast<f32x4*>(Ks + row * LDT + c4) = kv;
#pragma unroll
      for (int c = 0; c < 4; ++c) Vt[(c4 + c) * LDT + row] = vv[c];
    }
    __syncthreads();

    // S^T tiles: s[kt][e] = score(query iq, key j0 + 16kt + 4g + e)
    f32x4 s[4];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      s[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const f32x4 kf = *reinterpret_cast<const f32x4*>(Ks + (kt * 16 + r) * LDT + 16 * kk + 4 * g);
#pragma unroll
        for (int e = 0; e < 4; ++e) s[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[e], qf[kk][e], s[kt], 0, 0, 0);
      }
    }
    // mask, running max / sum over the query's 4 lanes
    float mt = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int j = j0 + kt * 16 + 4 * g + e;
        s[kt][e] = (j < lim) ? s[kt][e] * p.scale : -INFINITY;
        mt = fmaxf(mt, s[kt][e]);
      }
    mt = fmaxf(mt, __shfl_xor(mt, 16, 64));
    mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
    const float mn = fmaxf(m_run, mt);
    float corr = 1.f, ls = 0.f;
    if (mn > -INFINITY) {
      corr = (m_run > -INFINITY) ? expf(m_run - mn) : 0.f;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float pe = (s[kt][e] > -INFINITY) ? expf(s[kt][e] - mn) : 0.f;
          s[kt][e] = pe;
          ls += pe;
        }
    } else {
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) s[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    ls += __shfl_xor(ls, 16, 64);
    ls += __shfl_xor(ls, 32, 64);
    l_run = l_run * corr + ls;
    m_run = mn;
    // O^T = O^T corr + V^T . P^T -- the tile's product in a FRESH accumulator (a 64-key chain), then one add into the running sum:
    // accumulating straight into o made one chain over all Tk keys, whose rounding grew with the utterance length (round 6:
    // unit logits of an 11-s utterance sat 1.6x farther from float64 than the oracle's, of a 1.5-s one 0.8x)
    f32x4 ot[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) ot[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const f32x4 vf = *reinterpret_cast<const f32x4*>(Vt + (dt * 16 + r) * LDT + kt * 16 + 4 * g);
#pragma unroll
        for (int e = 0; e < 4; ++e) ot[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[e], s[kt][e], ot[dt], 0, 0, 0);
      }
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int e = 0; e < 4; ++e) o[dt][e] = o[dt][e] * corr + ot[dt][e];
  }
  if (q_ok) {
    const float inv = 1.0f / l_run;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      f32x4 v = o[dt];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] *= inv;
      *reinterpret_cast<f32x4*>(p.O + (size_t)iq * p.ldo + hoff + 16 * dt + 4 * g) = v;
    }
  }
}

// -------------------------------------------------------------------------------------------------
// MFMA form of the Conformer's relative-position attention (espnet_multihead_attention.py:154-209):
//   score[i,j] = ((q_i+u).k_j + (q_i+v).p[j-i+T-1]) / 8, chunk mask, softmax, PV.
// AC = K.(Q+u)^T and PV are the tiles of attention_mfma_kernel.  BD is Toeplitz in (i, j): for a wave's
// 16 queries and a 16-key sub-tile only 31 table rows can occur, so G^T = Pwin.(Q+v)^T is computed for
// the 32-row window starting at the sub-tile's smallest offset (two more MFMA tiles) and the skewed
// read BD[i][j] = G[i][15 - (i - ib) + (j - jb)] goes through a 2.3 KB per-wave LDS patch (the
// rel_shift of the reference, never materialised beyond 16 x 32).  The projected table rows a
// (64 query x 64 key) block can touch (127) are staged in LDS next to the K and V^T tiles.
// -------------------------------------------------------------------------------------------------
constexpr int PWIN = 2 * KT;          // staged table rows (127 used, last one zero)
constexpr int LDG = 36;               // per-wave G patch row stride
constexpr size_t kRelposLds = (size_t)(KT * LDT + DH * LDT + PWIN * LDT + 4 * 16 * LDG) * sizeof(float);

//
// SPLIT (single utterance, few query tiles): the key tiles are cut into p.ksplit groups of p.ktiles_per_split, one workgroup
// per (query tile, group, head) -- blockIdx.x = group * query_tiles + query_tile.  Each parks its un-normalised (o, m, l) in
// its register layout (5 b128 per thread, agent-scope sc1 stores), then bumps the (query tile, head) counter; the workgroup
// whose bump completes the count merges ALL groups in group order (so the result does not depend on who merges), writes O
// and zeroes the counter.  Same hand-off rules as conv_sk2: sc1 b128 stores -> vmcnt(0) -> barrier -> relaxed agent-scope
// atomic; sc1 loads on the other side (MI355X_MICROARCH.md: per-XCD L2s are not coherent).  Nobody waits for anybody.
template <bool SPLIT>
__global__ __launch_bounds__(256) void attention_relpos_mfma_kernel(AttnArgs p) {
  if (!SPLIT && p.nseg > 0) {
    const int* sg = p.segs + 4 * blockIdx.z;
    p.Tq = sg[1]; p.Tk = sg[3];
    if ((int)blockIdx.x * MQ >= p.Tq) return;
    p.Q += (size_t)sg[0] * p.ldq; p.O += (size_t)sg[0] * p.ldo;
    p.K += (size_t)sg[2] * p.ldk; p.V += (size_t)sg[2] * p.ldv;
    p.P += (size_t)(p.p_tmax - p.Tk) * p.ldp;
  }
  extern __shared__ __attribute__((aligned(16))) float smem_rp[];
  float* Ks = smem_rp;                  // [key][d]
  float* Vt = Ks + KT * LDT;            // [d][key]
  float* Ps = Vt + DH * LDT;            // [table row][d]
  using f32x4 = __attribute__((ext_vector_type(4))) float;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  float* Gs = Ps + PWIN * LDT + wave * 16 * LDG;
  const int r = lane & 15, g = lane >> 4;
  const int h = blockIdx.y, hoff = h * DH;
  int qt = blockIdx.x, sp = 0;
  if (SPLIT) { const int nqt = gridDim.x / p.ksplit; sp = blockIdx.x / nqt; qt = blockIdx.x - sp * nqt; }
  const int i0 = qt * MQ;
  const int q0 = p.q0;
  const int iq = i0 + wave * 16 + r;                 // this lane's query row (relative to Q); absolute position q0 + iq
  const bool q_ok = iq < p.Tq;

  f32x4 quf[4], qvf[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    f32x4 q = {0.f, 0.f, 0.f, 0.f};
    if (q_ok) q = *reinterpret_cast<const f32x4*>(p.Q + (size_t)iq * p.ldq + hoff + 16 * kk + 4 * g);
    const f32x4 bu = *reinterpret_cast<const f32x4*>(p.bias_u + hoff + 16 * kk + 4 * g);
    const f32x4 bv = *reinterpret_cast<const f32x4*>(p.bias_v + hoff + 16 * kk + 4 * g);
#pragma unroll
    for (int e = 0; e < 4; ++e) { quf[kk][e] = q[e] + bu[e]; qvf[kk][e] = q[e] + bv[e]; }
  }
  f32x4 o[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY, l_run = 0.f;

  int kmax = p.Tk;
  const int ilast = min(i0 + MQ, p.Tq) - 1;
  if (p.chunk > 0) kmax = min(kmax, ((ilast + q0) / p.chunk + 1) * p.chunk);
  int lim = p.Tk;
  if (p.chunk > 0) lim = min(lim, ((iq + q0) / p.chunk + 1) * p.chunk);
  if (!q_ok) lim = 0;
  int jbeg = 0, jend = kmax, s_eff = 1;
  if (SPLIT) {
    const int span = p.ktiles_per_split * KT;
    s_eff = (kmax + span - 1) / span;                // groups that hold a visible key of this query tile (chunk mask)
    if (sp >= s_eff) return;
    jbeg = sp * span;
    jend = min(kmax, jbeg + span);
  }

  for (int j0 = jbeg; j0 < jend; j0 += KT) {
    __syncthreads();
    for (int f = t; f < KT * (DH / 4); f += 256) {
      const int row = f >> 4, c4 = (f & 15) * 4;
      const int j = j0 + row;
      f32x4 kv = {0.f, 0.f, 0.f, 0.f}, vv = kv;
      if (j < p.Tk) {
        kv = *reinterpret_cast<const f32x4*>(p.K + (size_t)j * p.ldk + hoff + c4);
        vv = *reinterpret_cast<const f32x4*>(p.V + (size_t)j * p.ldv + hoff + c4);
      }
      *reinterpret_cast<f32x4*>(Ks + row * LDT + c4) = kv;
#pragma unroll
      for (int c = 0; c < 4; ++c) Vt[(c4 + c) * LDT + row] = vv[c];
    }
    {
      // local row lr <-> table row pbase + lr, pbase = j0 - (q0 + i0 + MQ - 1) + Tk - 1
      const int pbase = j0 - (q0 + i0 + MQ - 1) + p.Tk - 1;
      for (int f = t; f < PWIN * (DH / 4); f += 256) {
        const int row = f >> 4, c4 = (f & 15) * 4;
        const int pr = pbase + row;
        f32x4 pv = {0.f, 0.f, 0.f, 0.f};
        if (row < PWIN - 1 && pr >= 0 && pr < 2 * p.Tk - 1)
          pv = *reinterpret_cast<const f32x4*>(p.P + (size_t)pr * p.ldp + hoff + c4);
        *reinterpret_cast<f32x4*>(Ps + row * LDT + c4) = pv;
      }
    }
    __syncthreads();

    f32x4 s[4];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      s[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
      f32x4 ga = {0.f, 0.f, 0.f, 0.f}, gb = {0.f, 0.f, 0.f, 0.f};
      const int lrow0 = 16 * (kt - wave) + (MQ - 16);          // first window row of this (wave, sub-tile)
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const f32x4 kf = *reinterpret_cast<const f32x4*>(Ks + (kt * 16 + r) * LDT + 16 * kk + 4 * g);
        const f32x4 pa = *reinterpret_cast<const f32x4*>(Ps + (lrow0 + r) * LDT + 16 * kk + 4 * g);
        const f32x4 pb = *reinterpret_cast<const f32x4*>(Ps + (lrow0 + 16 + r) * LDT + 16 * kk + 4 * g);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          s[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[e], quf[kk][e], s[kt], 0, 0, 0);
          ga = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[e], qvf[kk][e], ga, 0, 0, 0);
          gb = __builtin_amdgcn_mfma_f32_16x16x4f32(pb[e], qvf[kk][e], gb, 0, 0, 0);
        }
      }
      // G[query r][window offset x]: x = 4g+e in ga, 16+4g+e in gb; BD for key 4g+e is x = 15 - r + 4g + e
      *reinterpret_cast<f32x4*>(Gs + r * LDG + 4 * g) = ga;
      *reinterpret_cast<f32x4*>(Gs + r * LDG + 16 + 4 * g) = gb;
      const float* gr = Gs + r * LDG + 15 - r + 4 * g;
#pragma unroll
      for (int e = 0; e < 4; ++e) s[kt][e] += gr[e];
    }
    float mt = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int j = j0 + kt * 16 + 4 * g + e;
        s[kt][e] = (j < lim) ? s[kt][e] * p.scale : -INFINITY;
        mt = fmaxf(mt, s[kt][e]);
      }
    mt = fmaxf(mt, __shfl_xor(mt, 16, 64));
    mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
    const float mn = fmaxf(m_run, mt);
    float corr = 1.f, ls = 0.f;
    if (mn > -INFINITY) {
      corr = (m_run > -INFINITY) ? expf(m_run - mn) : 0.f;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float pe = (s[kt][e] > -INFINITY) ? expf(s[kt][e] - mn) : 0.f;
          s[kt][e] = pe;
          ls += pe;
        }
    } else {
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) s[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    ls += __shfl_xor(ls, 16, 64);
    ls += __shfl_xor(ls, 32, 64);
    l_run = l_run * corr + ls;
    m_run = mn;
    // the tile's P V product in a fresh accumulator, then one add into the running sum (see attention_mfma_kernel)
    f32x4 ot[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) ot[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const f32x4 vf = *reinterpret_cast<const f32x4*>(Vt + (dt * 16 + r) * LDT + kt * 16 + 4 * g);
#pragma unroll
        for (int e = 0; e < 4; ++e) ot[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[e], s[kt][e], ot[dt], 0, 0, 0);
      }
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int e = 0; e < 4; ++e) o[dt][e] = o[dt][e] * corr + ot[dt][e];
  }
  if (SPLIT && s_eff > 1) {
    using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
    const int slot0 = (qt * (int)gridDim.y + h) * p.ksplit;
    {
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
          (void*)(p.part + (size_t)(slot0 + sp) * ATTN_PART_FLOATS), 0, ATTN_PART_FLOATS * 4, 0x00020000);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        u32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = __float_as_uint(o[dt][e]);
        __builtin_amdgcn_raw_buffer_store_b128(v, rs, (dt * 256 + t) * 16, 0, 16);      // aux 16 = sc1 (agent scope)
      }
      const u32x4 ml = {__float_as_uint(m_run), __float_as_uint(l_run), 0u, 0u};
      __builtin_amdgcn_raw_buffer_store_b128(ml, rs, (4 * 256 + t) * 16, 0, 16);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    __shared__ int s_last;
    unsigned* cnt = p.cnt + qt * (int)gridDim.y + h;
    if (t == 0) s_last = (__hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(s_eff - 1)) ? 1 : 0;
    __syncthreads();
    if (!s_last) return;
    // ---- the last arrival merges every group, in group order ----
    float mx = -INFINITY;
    for (int g2 = 0; g2 < s_eff; ++g2) {
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
          (void*)(p.part + (size_t)(slot0 + g2) * ATTN_PART_FLOATS), 0, ATTN_PART_FLOATS * 4, 0x00020000);
      const u32x4 ml = __builtin_amdgcn_raw_buffer_load_b128(rs, (4 * 256 + t) * 16, 0, 16);
      mx = fmaxf(mx, __uint_as_float(ml[0]));
    }
    float lsum = 0.f;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int g2 = 0; g2 < s_eff; ++g2) {
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
          (void*)(p.part + (size_t)(slot0 + g2) * ATTN_PART_FLOATS), 0, ATTN_PART_FLOATS * 4, 0x00020000);
      u32x4 v[5];
#pragma unroll
      for (int dt = 0; dt < 5; ++dt) v[dt] = __builtin_amdgcn_raw_buffer_load_b128(rs, (dt * 256 + t) * 16, 0, 16);
      const float mg = __uint_as_float(v[4][0]);
      const float f = (mg > -INFINITY) ? expf(mg - mx) : 0.f;
      lsum += __uint_as_float(v[4][1]) * f;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int e = 0; e < 4; ++e) o[dt][e] += __uint_as_float(v[dt][e]) * f;
    }
    l_run = lsum;
    if (t == 0) __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch of this context
  }
  if (q_ok) {
    const float inv = 1.0f / l_run;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      f32x4 v = o[dt];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] *= inv;
      *reinterpret_cast<f32x4*>(p.O + (size_t)iq * p.ldo + hoff + 16 * dt + 4 * g) = v;
    }
  }
}

// -------------------------------------------------------------------------------------------------
// Rel-pos attention of a FEW query rows over all keys: the tail rows of the incremental streaming encoder (<= 48 per policy() call,
// 8-16 for the 320-ms agent; ss_encoder_stream_forward).  attention_relpos_mfma_kernel gives such a call one workgroup per (head, key
// split) whose four waves are four 16-query sub-tiles -- three of them empty -- and walks a 64-key tile through LDS (K, V^T and 127
// table rows staged, 192 MFMAs per wave and tile): 14 us per layer, all latency.  Here a workgroup is (16-query tile, 64-key tile,
// head) and its four waves are the four 16-KEY sub-tiles: every operand fragment (q + u, q + v, K rows, the 31 table rows the sub-tile
// can touch, V columns) is loaded straight into registers in ONE round trip, a wave issues 48 + 16 MFMAs, and the waves meet twice in
// LDS (softmax statistics, then the 16 x 64 output sum).  Same arithmetic per score as the tile kernel (the BD term through the per-wave
// skew patch); the softmax is normalised per key tile and merged in key-tile order by the last arriving workgroup of a (query tile,
// head) -- the hand-off protocol of the SPLIT form above.
// -------------------------------------------------------------------------------------------------
constexpr int Q16_LDO = 68;           // padded row of the cross-wave output sum

__global__ __launch_bounds__(256) void attention_relpos_q16_kernel(AttnArgs p) {
  using f32x4 = __attribute__((ext_vector_type(4))) float;
  using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
  __shared__ __attribute__((aligned(16))) float Gs_all[4 * 16 * LDG];
  __shared__ float ms[4][16], ls[4][16];
  __shared__ __attribute__((aligned(16))) float osum[4][16][Q16_LDO];
  __shared__ int s_last;
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int r = lane & 15, g = lane >> 4;
  const int h = blockIdx.y, hoff = h * DH;
  const int nqt = gridDim.x / p.ksplit;
  const int sp = blockIdx.x / nqt, qt = blockIdx.x - sp * nqt;
  const int i0 = qt * 16, q0 = p.q0;
  const int ilast = min(i0 + 16, p.Tq) - 1;
  int kmax = p.Tk;
  if (p.chunk > 0) kmax = min(kmax, ((ilast + q0) / p.chunk + 1) * p.chunk);
  const int s_eff = (kmax + KT - 1) / KT;              // key tiles that hold a visible key of this query tile
  if (sp >= s_eff) return;
  const int jb = sp * KT + 16 * wave;                   // this wave's 16 keys
  const int iq = i0 + r;
  const bool q_ok = iq < p.Tq;
  int lim = p.Tk;
  if (p.chunk > 0) lim = min(lim, ((iq + q0) / p.chunk + 1) * p.chunk);
  if (!q_ok) lim = 0;

  // ---- every operand of the wave in one round trip ----
  f32x4 quf[4], qvf[4], kf[4], pa[4], pb[4];
  float vv[4][4];
  const int pbase = jb - (q0 + i0 + 15) + p.Tk - 1;     // table row of window row 0: key jb against the tile's last query
  const int pra = pbase + r, prb = pbase + 16 + r;
  const bool pa_ok = pra >= 0 && pra < 2 * p.Tk - 1, pb_ok = r < 15 && prb >= 0 && prb < 2 * p.Tk - 1;
  const bool k_ok = jb + r < p.Tk;
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    const int c = hoff + 16 * kk + 4 * g;
    f32x4 q = {0.f, 0.f, 0.f, 0.f};
    if (q_ok) q = *reinterpret_cast<const f32x4*>(p.Q + (size_t)iq * p.ldq + c);
    const f32x4 bu = *reinterpret_cast<const f32x4*>(p.bias_u + c);
    const f32x4 bv = *reinterpret_cast<const f32x4*>(p.bias_v + c);
#pragma unroll
    for (int e = 0; e < 4; ++e) { quf[kk][e] = q[e] + bu[e]; qvf[kk][e] = q[e] + bv[e]; }
    kf[kk] = pa[kk] = pb[kk] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (k_ok) kf[kk] = *reinterpret_cast<const f32x4*>(p.K + (size_t)(jb + r) * p.ldk + c);
    if (pa_ok) pa[kk] = *reinterpret_cast<const f32x4*>(p.P + (size_t)pra * p.ldp + c);
    if (pb_ok) pb[kk] = *reinterpret_cast<const f32x4*>(p.P + (size_t)prb * p.ldp + c);
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int key = jb + 4 * g + e;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) vv[dt][e] = key < p.Tk ? p.V[(size_t)key * p.ldv + hoff + dt * 16 + r] : 0.f;
  }

  // ---- scores: AC = K (q + u)^T, BD through the skew patch (see attention_relpos_mfma_kernel) ----
  f32x4 s = {0.f, 0.f, 0.f, 0.f}, ga = s, gb = s;
#pragma unroll
  for (int kk = 0; kk < 4; ++kk)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      s = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[kk][e], quf[kk][e], s, 0, 0, 0);
      ga = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[kk][e], qvf[kk][e], ga, 0, 0, 0);
      gb = __builtin_amdgcn_mfma_f32_16x16x4f32(pb[kk][e], qvf[kk][e], gb, 0, 0, 0);
    }
  float* Gs = Gs_all + wave * 16 * LDG;
  *reinterpret_cast<f32x4*>(Gs + r * LDG + 4 * g) = ga;
  *reinterpret_cast<f32x4*>(Gs + r * LDG + 16 + 4 * g) = gb;
  const float* gr = Gs + r * LDG + 15 - r + 4 * g;
  float mw = -INFINITY;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int j = jb + 4 * g + e;
    s[e] = (j < lim) ? (s[e] + gr[e]) * p.scale : -INFINITY;
    mw = fmaxf(mw, s[e]);
  }
  mw = fmaxf(mw, __shfl_xor(mw, 16, 64));
  mw = fmaxf(mw, __shfl_xor(mw, 32, 64));
  float pe[4], lw = 0.f;
#pragma unroll
  for (int e = 0; e < 4; ++e) { pe[e] = (s[e] > -INFINITY) ? expf(s[e] - mw) : 0.f; lw += pe[e]; }
  lw += __shfl_xor(lw, 16, 64);
  lw += __shfl_xor(lw, 32, 64);
  if (g == 0) { ms[wave][r] = mw; ls[wave][r] = lw; }
  __syncthreads();
  {
    const float mt = fmaxf(fmaxf(ms[0][r], ms[1][r]), fmaxf(ms[2][r], ms[3][r]));
    const float fw = (mw > -INFINITY) ? expf(mw - mt) : 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) pe[e] *= fw;
  }
  // ---- P V: D[query][d] += P[query][key] V[key][d] -- lane (r, g) ends up with queries 4 g + e', head column 16 dt + r ----
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) {
    f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 4; ++e) o = __builtin_amdgcn_mfma_f32_16x16x4f32(pe[e], vv[dt][e], o, 0, 0, 0);
#pragma unroll
    for (int e = 0; e < 4; ++e) osum[wave][4 * g + e][dt * 16 + r] = o[e];
  }
  __syncthreads();
  // ---- thread (query q, columns d4 .. d4 + 3): the four waves' sums in wave order, the tile's (m, l) ----
  const int q = t >> 4, d4 = (t & 15) * 4;
  f32x4 acc = *reinterpret_cast<const f32x4*>(&osum[0][q][d4]);
#pragma unroll
  for (int w2 = 1; w2 < 4; ++w2) {
    const f32x4 o2 = *reinterpret_cast<const f32x4*>(&osum[w2][q][d4]);
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[e] += o2[e];
  }
  float m_run = fmaxf(fmaxf(ms[0][q], ms[1][q]), fmaxf(ms[2][q], ms[3][q])), l_run = 0.f;
#pragma unroll
  for (int w2 = 0; w2 < 4; ++w2) l_run += (ms[w2][q] > -INFINITY) ? ls[w2][q] * expf(ms[w2][q] - m_run) : 0.f;

  if (s_eff > 1) {
    const int slot0 = (qt * (int)gridDim.y + h) * p.ksplit;
    {
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
          (void*)(p.part + (size_t)(slot0 + sp) * ATTN_PART_FLOATS), 0, ATTN_PART_FLOATS * 4, 0x00020000);
      const u32x4 v = {__float_as_uint(acc[0]), __float_as_uint(acc[1]), __float_as_uint(acc[2]), __float_as_uint(acc[3])};
      __builtin_amdgcn_raw_buffer_store_b128(v, rs, (q * 64 + d4) * 4, 0, 16);          // aux 16 = sc1 (agent scope)
      if ((t & 15) == 0) {
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(m_run), rs, (1024 + q) * 4, 0, 16);
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(l_run), rs, (1040 + q) * 4, 0, 16);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    unsigned* cnt = p.cnt + qt * (int)gridDim.y + h;
    if (t == 0) s_last = (__hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(s_eff - 1)) ? 1 : 0;
    __syncthreads();
    if (!s_last) return;
    // ---- the last arrival merges every key tile, in key-tile order ----
    float mx = -INFINITY;
    for (int g2 = 0; g2 < s_eff; ++g2) {
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
          (void*)(p.part + (size_t)(slot0 + g2) * ATTN_PART_FLOATS), 0, ATTN_PART_FLOATS * 4, 0x00020000);
      mx = fmaxf(mx, __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, (1024 + q) * 4, 0, 16)));
    }
    acc = f32x4{0.f, 0.f, 0.f, 0.f};
    l_run = 0.f;
    for (int g2 = 0; g2 < s_eff; ++g2) {
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
          (void*)(p.part + (size_t)(slot0 + g2) * ATTN_PART_FLOATS), 0, ATTN_PART_FLOATS * 4, 0x00020000);
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (q * 64 + d4) * 4, 0, 16);
      const float mg = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, (1024 + q) * 4, 0, 16));
      const float lg = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, (1040 + q) * 4, 0, 16));
      const float f = (mg > -INFINITY) ? expf(mg - mx) : 0.f;
      l_run += lg * f;
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e] += __uint_as_float(v[e]) * f;
    }
    if (t == 0) __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch of this context
  }
  if (i0 + q < p.Tq) {
    const float inv = 1.0f / l_run;
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[e] *= inv;
    *reinterpret_cast<f32x4*>(p.O + (size_t)(i0 + q) * p.ldo + hoff + d4) = acc;
  }
}

int launch_attention(const AttnArgs& a, hipStream_t stream) {
  const int tq = a.nseg > 0 ? a.max_q : a.Tq;
  if (tq <= 0 || (a.nseg == 0 && a.Tk <= 0)) return SS_OK;
  if ((a.ldk & 3) || (a.ldv & 3)) return SS_ERR_ARG;
  const int gz = a.nseg > 0 ? a.nseg : 1;
  if (a.q0 != 0 && (a.nseg > 0 || a.causal)) return SS_ERR_ARG;
  if (!a.P && tq <= 8 && a.q0 == 0 && !a.no_decode_kernel) {
    hipLaunchKernelGGL(attention_decode_kernel, dim3(tq, a.H, gz), dim3(256), 0, stream, a);
    SS_LAUNCH_CHECK();
    return SS_OK;
  }
  if (!a.P && a.q0 == 0 && !disp().attn_no_mfma && ((a.ldq | a.ldo) & 3) == 0) {
    hipLaunchKernelGGL(attention_mfma_kernel, dim3(cdiv(tq, MQ), a.H, gz), dim3(256), 0, stream, a);
    SS_LAUNCH_CHECK();
    return SS_OK;
  }
  dim3 grid(cdiv(tq, QB), a.H, gz);
  if (a.P) {
    if ((a.nseg == 0 && a.q0 + a.Tq != a.Tk) || (a.ldp & 3) || !a.bias_u || !a.bias_v) return SS_ERR_ARG;
    if (a.nseg > 0 && a.p_tmax <= 0) return SS_ERR_ARG;
    if (!disp().attn_no_mfma && ((a.ldq | a.ldo) & 3) == 0 && a.k_mask_tail == 0 && !a.causal) {
      const int nqt = cdiv(tq, MQ), nkt = cdiv(a.Tk, KT);
      // a few query rows over all keys (the incremental streaming encoder's tail rows): 16-query tiles, one workgroup per key tile
      if (a.nseg == 0 && tq <= 48 && disp().attn_q16 && (nkt == 1 || (a.part && cdiv(tq, 16) * a.H <= a.cnt_slots && cdiv(tq, 16) * a.H * nkt <= a.part_slots))) {
        AttnArgs b = a;
        b.ksplit = nkt; b.ktiles_per_split = 1;
        hipLaunchKernelGGL(attention_relpos_q16_kernel, dim3(cdiv(tq, 16) * nkt, a.H, 1), dim3(256), 0, stream, b);
        SS_LAUNCH_CHECK();
        return SS_OK;
      }
      // key split: one utterance with few (query tile, head) pairs and more than one key tile
      if (a.nseg == 0 && a.part && disp().attn_split >= 0 && nkt >= 2 && nqt * a.H <= a.cnt_slots && nqt * a.H < 128) {
        int tps = disp().attn_split > 0 ? disp().attn_split : cdiv(nkt, std::min(16, std::max(1, 256 / (nqt * a.H))));
        tps = std::max(tps, cdiv(nkt, 16));
        const int S = cdiv(nkt, tps);
        if (S >= 2 && nqt * a.H * S <= a.part_slots) {
          AttnArgs b = a;
          b.ksplit = S; b.ktiles_per_split = tps;
          SS_MAX_LDS_ONCE(&attention_relpos_mfma_kernel<true>, kRelposLds);
          hipLaunchKernelGGL(attention_relpos_mfma_kernel<true>, dim3(nqt * S, a.H, 1), dim3(256), kRelposLds, stream, b);
          SS_LAUNCH_CHECK();
          return SS_OK;
        }
      }
      SS_MAX_LDS_ONCE(&attention_relpos_mfma_kernel<false>, kRelposLds);
      hipLaunchKernelGGL(attention_relpos_mfma_kernel<false>, dim3(nqt, a.H, gz), dim3(256), kRelposLds, stream, a);
      SS_LAUNCH_CHECK();
      return SS_OK;
    }
    hipLaunchKernelGGL(attention_kernel<true>, grid, dim3(256), 0, stream, a);
  } else {
    hipLaunchKernelGGL(attention_kernel<false>, grid, dim3(256), 0, stream, a);
  }
  SS_LAUNCH_CHECK();
  return SS_OK;
}

}  // namespace ss
