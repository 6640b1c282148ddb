// One greedy decode step of the MT text decoder (one new token, B = 1) as ONE persistent launch
// (reference: agent/sequence_generator.py:165-582 at beam 1 over researches/ctc_unity/modules/transformer_decoder.py:257-403 and
// transformer_layer.py:388-551).
//
// The launch-per-op form of the step is a chain of 35 dependent kernels at ~5 us each: every op needs the WHOLE
// output vector of the previous one, so nothing fuses without a grid-wide exchange.  Here the exchange happens inside the launch:
// G resident workgroups run the step's 34 phases; a phase's output vector (512 ... 2112 floats) is published as 8-byte {epoch, value}
// granules with relaxed agent-scope 64-bit stores -- the data is the flag (cdna_hip_programming.md Guideline 16, form R2: no fences,
// no barrier counters, placement-independent) -- and every workgroup of the next phase sweeps the granules into its LDS until all
// tags carry this launch's epoch.  A phase's weight rows are requested BEFORE the sweep, so the weight stream hides under the
// exchange latency (tools/src/allgather_probe.hip: 3.3-4.0 us per phase with 2 MB of weights, against ~5.2 us per kernel).
//   per layer:  A  qkv = Wqkv LN1(x)            (publishes 1536; the wave that owns a k / v column also writes the KV cache row)
//               B  self-attention, one workgroup per head over the cache + the new key        (512)
//               C  x1 = x + Wo a                                                              (512)
//               D  q2 = Wcq LN2(x1)                                                           (512)
//               E  cross-attention, 8 heads x 4 key ranges, partial (m, l, acc[64])           (2112)
//               F  x2 = x1 + Wco merge(partials)                                              (512)
//               G  h = relu(W1 LN3(x2))                                                       (2048)
//               H  x = x2 + W2 h                                                              (512)
//   then        I  feats = LN_f(x) (workgroup 0 writes the row), logits = E feats, masked arg-max per workgroup   (2 G)
//               J  workgroup 0 reduces the candidates (first maximum wins, as torch.max) and writes the next token.
// Every spin is bounded: a time-out bumps the context's own error word (collected into ss_debug_sk_errors by the host) and the launch runs to its
// end on whatever it has -- results are then wrong and the counter says so; nothing hangs.  The granule region is zeroed once
// at allocation and the epoch grows with every launch of the context, so no per-launch memset is needed.
// Opt-in per context (ss_mt_set_persistent(m, G), or SS_MT_PERSISTENT=G in the environment): all G workgroups of a launch must become resident, which is certain
// for one decoding stream and not when many contexts decode at once next to full-chip kernels (each would hold CUs while
// waiting for its missing workgroups); the default path stays the launch-per-op one.
#include "mt_step.hpp"

namespace ss {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
#define MT_RLX __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

namespace {

[[maybe_unused]] __device__ __forceinline__ float mt_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
[[maybe_unused]] __device__ __forceinline__ float mt_wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
[[maybe_unused]] __device__ __forceinline__ float mt_rdlane(float v, int l) { return __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), l)); }

[[maybe_unused]] __device__ __forceinline__ void mt_publish(mt_u64* g, unsigned epoch, float v) {
  __hip_atomic_store(g, ((mt_u64)epoch << 32) | __float_as_uint(v), MT_RLX);
}

// All 256 threads sweep N granules (N <= 2304) into dst until every tag carries `epoch`; bounded.
[[maybe_unused]] __device__ __forceinline__ void mt_gather(const mt_u64* g, int N, float* dst, unsigned epoch, unsigned* err, int t, int lane) {
  constexpr int MAXL = 9;
  unsigned long long t_first = 0ull;     // wall clock (100 MHz) of the first sweep that found a granule missing
  __syncthreads();                       // every wave is done reading what dst held before
  for (;;) {
    mt_u64 x[MAXL];
#pragma unroll
    for (int k = 0; k < MAXL; ++k) {
      const int idx = t + k * 256;
      x[k] = (mt_u64)epoch << 32;
      if (idx < N) x[k] = __hip_atomic_load(g + idx, MT_RLX);
    }
    bool ok = true;
#pragma unroll
    for (int k = 0; k < MAXL; ++k) {
      const int idx = t + k * 256;
      ok &= (unsigned)(x[k] >> 32) == epoch;
      if (idx < N) dst[idx] = __uint_as_float((unsigned)x[k]);
    }
    if (__all(ok)) break;
    // bounded by TIME (2^18 sweeps were hundreds of ms per wave): a granule that has not arrived MT_WAIT_TICKS after
    // it was first missed means a workgroup of this launch is not resident -- count it, let the launch run out, the host falls back
    const unsigned long long now = wall_clock64();
    if (t_first == 0ull) t_first = now;
    const bool late = now - t_first > MT_WAIT_TICKS;
    if (late || __hip_atomic_load(err, MT_RLX) != 0u) {
      if (lane == 0 && late) atomicAdd(err, 1u);
      break;
    }
    __builtin_amdgcn_s_sleep(1);
  }
  __syncthreads();
}

// y = LayerNorm(x) over 512 values in LDS (eps 1e-5, biased variance of deviations from the mean), all 256 threads.
[[maybe_unused]] __device__ __forceinline__ void mt_layernorm(const float* x, const float* __restrict__ gm, const float* __restrict__ bt, float* y,
                                             float* red, int t, int lane, int wave) {
  const float a = x[t], b = x[t + 256];
  float s = mt_wave_sum(a + b);
  if (lane == 0) red[wave] = s;
  __syncthreads();
  const float mean = ((red[0] + red[1]) + (red[2] + red[3])) * (1.0f / MT_D);
  const float da = a - mean, db = b - mean;
  float q = mt_wave_sum(da * da + db * db);
  if (lane == 0) red[4 + wave] = q;
  __syncthreads();
  const float rstd = 1.0f / sqrtf(((red[4] + red[5]) + (red[6] + red[7])) * (1.0f / MT_D) + 1e-5f);
  y[t] = da * rstd * gm[t] + bt[t];
  y[t + 256] = db * rstd * gm[t + 256] + bt[t + 256];
  __syncthreads();
}

// Weight rows of up to 8 columns (K = 512) of this wave: column c is gw + c * nw.
struct MtW512 { f32x4 w[8][2]; };
[[maybe_unused]] __device__ __forceinline__ void mt_fetch512(MtW512& r, const float* __restrict__ W, int N, int gw, int nw, int lane, int c0 = 0) {
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const int col = gw + (c0 + c) * nw;
    r.w[c][0] = f32x4{0.f, 0.f, 0.f, 0.f};
    r.w[c][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (col < N) {
      const float* row = W + (size_t)col * MT_D + 4 * lane;
      r.w[c][0] = *reinterpret_cast<const f32x4*>(row);
      r.w[c][1] = *reinterpret_cast<const f32x4*>(row + 256);
    }
  }
}
// dot products of the fetched columns with the 512-vector in LDS; lane c returns column c's sum (0 elsewhere)
[[maybe_unused]] __device__ __forceinline__ float mt_dot512(const MtW512& r, const float* in, int lane) {
  const f32x4 x0 = *reinterpret_cast<const f32x4*>(in + 4 * lane), x1 = *reinterpret_cast<const f32x4*>(in + 256 + 4 * lane);
  float mine = 0.f;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    float a0 = x0[0] * r.w[c][0][0], a1 = x0[1] * r.w[c][0][1], a2 = x0[2] * r.w[c][0][2], a3 = x0[3] * r.w[c][0][3];
    a0 = fmaf(x1[0], r.w[c][1][0], a0); a1 = fmaf(x1[1], r.w[c][1][1], a1);
    a2 = fmaf(x1[2], r.w[c][1][2], a2); a3 = fmaf(x1[3], r.w[c][1][3], a3);
    const float s = mt_wave_sum((a0 + a1) + (a2 + a3));
    if (lane == c) mine = s;
  }
  return mine;
}

}  // namespace

__global__ __launch_bounds__(256) void mt_step_kernel(const MtStepArgs p) {
#if __HIP_DEVICE_COMPILE__
  __shared__ __attribute__((aligned(16))) float xbuf[MT_D];      // residual stream
  __shared__ __attribute__((aligned(16))) float ybuf[MT_D];      // LayerNorm output / attention context
  __shared__ __attribute__((aligned(16))) float vec[2304];       // gathered vector of the phase
  __shared__ float red[16];
  __shared__ float part_m[4], part_l[4], part_acc[4][MT_DH];
  __shared__ float best_v[4];
  __shared__ int best_i[4];
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int G = gridDim.x, wg = blockIdx.x, nw = G * 4, gw = wg * 4 + wave;
  unsigned* err = p.err;
  __shared__ int s_tok;
  // Device-side token loop (round 4): one launch runs up to n_steps decode steps -- the token a step decides is the next step's
  // input, every workgroup works it out for itself from the exchanged arg-max candidates (no hop through workgroup 0, no host in
  // between) and the loop ends for all of them at </s>.  Step `it` uses epoch p.epoch + it; a phase's granules are reused every step:
  // a workgroup can only overwrite them after it has gathered the previous step's candidates, which every other workgroup publishes
  // after its last read of that step -- the all-to-all exchanges are the barrier.
  int tk = p.tok[0];
  // the prefix pass (launch-per-op, fed by ss_mt_greedy before this launch) may already have produced </s>: the search is over --
  // do not feed it and decode on until a later step happens to emit </s> again (~125 us per step on the streaming latency path,
  // and cache / feature rows past the end).  Uniform over workgroups: nobody waits for anybody.
  // Only the search loop of ss_mt_greedy (p.search): a single step of ss_mt_append with the persistent step on computes what it is
  // fed, </s> at position > 0 included, exactly like the launch-per-op form of the same call.
  if (p.search && p.pos0 > 0 && tk == p.eos) return;
#pragma unroll 1
  for (int it = 0; it < p.n_steps; ++it) {
  const unsigned epoch = p.epoch + (unsigned)it;
  const int pos0 = p.pos0 + it;
  const bool ban_eos = pos0 < p.min_len, force_eos = pos0 >= p.max_len;

  // ---- embedding: x0 = sqrt(D) E[tok] + sinusoid(position); a <pad> token takes position padding_idx (make_positions) ----
  {
    if ((unsigned)tk >= (unsigned)p.V) tk = p.eos;         // the chained token of a step that timed out (phase J: -1): stay in range
    const int pos = (tk == p.pad) ? p.pad : pos0 + p.pad + 1;
    xbuf[t] = p.emb_scale * p.emb[(size_t)tk * MT_D + t] + p.pos_table[(size_t)pos * MT_D + t];
    xbuf[t + 256] = p.emb_scale * p.emb[(size_t)tk * MT_D + t + 256] + p.pos_table[(size_t)pos * MT_D + t + 256];
  }
  __syncthreads();

  MtW512 wr;
#pragma unroll 1
  for (int l = 0; l < MT_L; ++l) {
    const MtLayerW& Lw = p.L[l];
    mt_u64* gl = p.gran + (size_t)l * MG_LAYER;
    // ================= A: qkv = Wqkv LN1(x) + b (q rows pre-scaled at pack time) =================
    mt_fetch512(wr, Lw.wqkv, 3 * MT_D, gw, nw, lane);
    if (l > 0) {                                           // x of this layer = phase H of the previous one
      mt_gather(p.gran + (size_t)(l - 1) * MG_LAYER + MG_X, MT_D, xbuf, epoch, err, t, lane);
    }
    mt_layernorm(xbuf, Lw.ln1_g, Lw.ln1_b, ybuf, red, t, lane, wave);
    {
      const float s = mt_dot512(wr, ybuf, lane);
      const int col = gw + lane * nw;
      if (lane < 8 && col < 3 * MT_D) {
        const float v = s + Lw.bqkv[col];
        mt_publish(gl + MG_QKV + col, epoch, v);
        // cache row of this position: read by later launches AND by later steps of this one (other CUs): write-through store
        __hip_atomic_store(Lw.selfbuf + (size_t)pos0 * 3 * MT_D + col, v, MT_RLX);
      }
    }
    // ================= B: causal self-attention over the cache rows 0 .. pos0-1 and the new key =================
    mt_fetch512(wr, Lw.wo, MT_D, gw, nw, lane);            // C's weights, under the exchange
    mt_gather(gl + MG_QKV, 3 * MT_D, vec, epoch, err, t, lane);
    if (wg < MT_H) {
      const int h = wg, hoff = h * MT_DH;
      const int kmax = pos0 + 1;
      const float* q = vec + hoff;                         // LDS
      const float* Kc = Lw.selfbuf + MT_D + hoff;          // rows j < pos0 (written by earlier launches)
      const float* Vc = Lw.selfbuf + 2 * MT_D + hoff;
      float m_run = -INFINITY, l_run = 0.f, acc = 0.f;
      for (int j0 = wave * 64; j0 < kmax; j0 += 256) {
        const int j = j0 + lane;
        const bool vis = j < kmax;
        float vv[64];
#pragma unroll
        for (int u = 0; u < 64; ++u) {
          const int jj = min(j0 + u, kmax - 1);
          vv[u] = jj < pos0 ? __hip_atomic_load(Vc + (size_t)jj * 3 * MT_D + lane, MT_RLX) : vec[2 * MT_D + hoff + lane];   // (L1-bypassing: rows of earlier steps of this launch)
        }
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        if (vis) {
          if (j < pos0) {
            // (ONE wave-uniform resource over the cache, the key row in the per-lane offset: a per-lane base pointer in the resource
            //  would make every load a 64-trip waterfall loop)
            const __amdgpu_buffer_rsrc_t rsK = __builtin_amdgcn_make_buffer_rsrc((void*)Kc, 0, 0x7ffffff0, 0x00020000);
            const int kofs = j * 3 * MT_D * 4;
#pragma unroll
            for (int d4 = 0; d4 < MT_DH / 4; ++d4) {
              const u32x4 kb = __builtin_amdgcn_raw_buffer_load_b128(rsK, kofs + d4 * 16, 0, 16);      // sc1: the row may come from an earlier step of this launch
              const float kx = __uint_as_float(kb[0]), ky = __uint_as_float(kb[1]), kz = __uint_as_float(kb[2]), kw = __uint_as_float(kb[3]);
              s0 = fmaf(q[4 * d4 + 0], kx, s0); s1 = fmaf(q[4 * d4 + 1], ky, s1);
              s2 = fmaf(q[4 * d4 + 2], kz, s2); s3 = fmaf(q[4 * d4 + 3], kw, s3);
            }
          } else {
            const float* kn = vec + MT_D + hoff;
#pragma unroll
            for (int d4 = 0; d4 < MT_DH / 4; ++d4) {
              s0 = fmaf(q[4 * d4 + 0], kn[4 * d4 + 0], s0); s1 = fmaf(q[4 * d4 + 1], kn[4 * d4 + 1], s1);
              s2 = fmaf(q[4 * d4 + 2], kn[4 * d4 + 2], s2); s3 = fmaf(q[4 * d4 + 3], kn[4 * d4 + 3], s3);
            }
          }
        }
        const float sv = vis ? ((s0 + s1) + (s2 + s3)) : -INFINITY;
        const float mn = fmaxf(m_run, mt_wave_max(sv));
        const float pe = vis ? expf(sv - mn) : 0.f;
        const float corr = (m_run > -INFINITY) ? expf(m_run - mn) : 0.f;
        l_run = l_run * corr + mt_wave_sum(pe);
        acc *= corr;
        m_run = mn;
#pragma unroll
        for (int u = 0; u < 64; ++u) acc = fmaf(mt_rdlane(pe, u), vv[u], acc);
      }
      if (lane == 0) { part_m[wave] = m_run; part_l[wave] = l_run; }
      part_acc[wave][lane] = acc;
      __syncthreads();
      if (wave == 0) {
        const float mt = fmaxf(fmaxf(part_m[0], part_m[1]), fmaxf(part_m[2], part_m[3]));
        float lsum = 0.f, o = 0.f;
#pragma unroll
        for (int w2 = 0; w2 < 4; ++w2) {
          const float f = (part_m[w2] > -INFINITY) ? expf(part_m[w2] - mt) : 0.f;
          lsum += part_l[w2] * f;
          o += part_acc[w2][lane] * f;
        }
        mt_publish(gl + MG_ATT + hoff + lane, epoch, o / lsum);
      }
    }
    // ================= C: x1 = x + Wo a + bo =================
    mt_gather(gl + MG_ATT, MT_D, ybuf, epoch, err, t, lane);
    {
      const float s = mt_dot512(wr, ybuf, lane);
      const int col = gw + lane * nw;
      if (lane < 8 && col < MT_D) mt_publish(gl + MG_X1 + col, epoch, (s + Lw.bo[col]) + xbuf[col]);
    }
    // ================= D: q2 = Wcq LN2(x1) + b =================
    mt_fetch512(wr, Lw.wcq, MT_D, gw, nw, lane);
    mt_gather(gl + MG_X1, MT_D, xbuf, epoch, err, t, lane);
    mt_layernorm(xbuf, Lw.ln2_g, Lw.ln2_b, ybuf, red, t, lane, wave);
    {
      const float s = mt_dot512(wr, ybuf, lane);
      const int col = gw + lane * nw;
      if (lane < 8 && col < MT_D) mt_publish(gl + MG_Q2 + col, epoch, s + Lw.bcq[col]);
    }
    // ================= E: cross-attention partials: head = wg % 8, key range = wg / 8 of MT_SPLITS =================
    mt_fetch512(wr, Lw.wco, MT_D, gw, nw, lane);            // F's weights
    mt_gather(gl + MG_Q2, MT_D, vec, epoch, err, t, lane);
    if (wg < MT_H * MT_SPLITS) {
      const int h = wg % MT_H, sp = wg / MT_H, hoff = h * MT_DH;
      const int k_lo = (int)((long long)p.Tp * sp / MT_SPLITS), k_hi = (int)((long long)p.Tp * (sp + 1) / MT_SPLITS);
      const float* q = vec + hoff;
      const float* Kc = Lw.cross + hoff;
      const float* Vc = Lw.cross + MT_D + hoff;
      float m_run = -INFINITY, l_run = 0.f, acc = 0.f;
      for (int j0 = k_lo + wave * 64; j0 < k_hi; j0 += 256) {
        const int j = j0 + lane;
        const bool vis = j < k_hi;
        float vv[64];
#pragma unroll
        for (int u = 0; u < 64; ++u) vv[u] = Vc[(size_t)min(j0 + u, k_hi - 1) * 2 * MT_D + lane];
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        if (vis) {
          const float4* kr = reinterpret_cast<const float4*>(Kc + (size_t)j * 2 * MT_D);
#pragma unroll
          for (int d4 = 0; d4 < MT_DH / 4; ++d4) {
            const float4 kv = kr[d4];
            s0 = fmaf(q[4 * d4 + 0], kv.x, s0); s1 = fmaf(q[4 * d4 + 1], kv.y, s1);
            s2 = fmaf(q[4 * d4 + 2], kv.z, s2); s3 = fmaf(q[4 * d4 + 3], kv.w, s3);
          }
        }
        const float sv = vis ? ((s0 + s1) + (s2 + s3)) : -INFINITY;
        const float mn = fmaxf(m_run, mt_wave_max(sv));
        const float pe = vis ? expf(sv - mn) : 0.f;
        const float corr = (m_run > -INFINITY) ? expf(m_run - mn) : 0.f;
        l_run = l_run * corr + mt_wave_sum(pe);
        acc *= corr;
        m_run = mn;
#pragma unroll
        for (int u = 0; u < 64; ++u) acc = fmaf(mt_rdlane(pe, u), vv[u], acc);
      }
      if (lane == 0) { part_m[wave] = m_run; part_l[wave] = l_run; }
      part_acc[wave][lane] = acc;
      __syncthreads();
      if (wave == 0) {
        const float mt = fmaxf(fmaxf(part_m[0], part_m[1]), fmaxf(part_m[2], part_m[3]));
        float lsum = 0.f, o = 0.f;
#pragma unroll
        for (int w2 = 0; w2 < 4; ++w2) {
          const float f = (part_m[w2] > -INFINITY) ? expf(part_m[w2] - mt) : 0.f;
          lsum += part_l[w2] * f;
          o += part_acc[w2][lane] * f;
        }
        mt_u64* gp = gl + MG_PART + (h * MT_SPLITS + sp) * MT_PARTV;
        mt_publish(gp + 2 + lane, epoch, o);               // un-normalised: sum_j exp(s_j - mt) v_j over the range
        if (lane == 0) { mt_publish(gp, epoch, mt); mt_publish(gp + 1, epoch, lsum); }
      }
    }
    // ================= F: x2 = x1 + Wco merge(partials) + b =================
    mt_gather(gl + MG_PART, MT_PART, vec, epoch, err, t, lane);
    {
      // merge the key ranges of each head (fixed order): thread t -> dims t, t + 256
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int d = t + half * 256, h = d / MT_DH, dd = d % MT_DH;
        const float* ph = vec + h * MT_SPLITS * MT_PARTV;
        float mt = -INFINITY;
#pragma unroll
        for (int sp = 0; sp < MT_SPLITS; ++sp) mt = fmaxf(mt, ph[sp * MT_PARTV]);
        float lsum = 0.f, o = 0.f;
#pragma unroll
        for (int sp = 0; sp < MT_SPLITS; ++sp) {
          const float ms = ph[sp * MT_PARTV];
          const float f = (ms > -INFINITY) ? expf(ms - mt) : 0.f;
          lsum += ph[sp * MT_PARTV + 1] * f;
          o += ph[sp * MT_PARTV + 2 + dd] * f;
        }
        ybuf[d] = o / lsum;
      }
      __syncthreads();
      const float s = mt_dot512(wr, ybuf, lane);
      const int col = gw + lane * nw;
      if (lane < 8 && col < MT_D) mt_publish(gl + MG_X2 + col, epoch, (s + Lw.bco[col]) + xbuf[col]);
    }
    // ================= G: h = relu(W1 LN3(x2) + b1) =================
    mt_fetch512(wr, Lw.w1, MT_F, gw, nw, lane);
    mt_gather(gl + MG_X2, MT_D, xbuf, epoch, err, t, lane);
    mt_layernorm(xbuf, Lw.ln3_g, Lw.ln3_b, ybuf, red, t, lane, wave);
    {
      const float s = mt_dot512(wr, ybuf, lane);
      const int col = gw + lane * nw;
      if (lane < 8 && col < MT_F) mt_publish(gl + MG_HID + col, epoch, fmaxf(s + Lw.b1[col], 0.f));
    }
    // ================= H: x = x2 + W2 h + b2   (K = 2048: up to two columns per wave) =================
    {
      f32x4 w2r[2][8];
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const int col = gw + c * nw;
#pragma unroll
        for (int it = 0; it < 8; ++it) {
          w2r[c][it] = f32x4{0.f, 0.f, 0.f, 0.f};
          if (col < MT_D) w2r[c][it] = *reinterpret_cast<const f32x4*>(Lw.w2 + (size_t)col * MT_F + it * 256 + 4 * lane);
        }
      }
      mt_gather(gl + MG_HID, MT_F, vec, epoch, err, t, lane);
      float mine = 0.f;
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
        for (int it = 0; it < 8; ++it) {
          const f32x4 x = *reinterpret_cast<const f32x4*>(vec + it * 256 + 4 * lane);
          a0 = fmaf(x[0], w2r[c][it][0], a0); a1 = fmaf(x[1], w2r[c][it][1], a1);
          a2 = fmaf(x[2], w2r[c][it][2], a2); a3 = fmaf(x[3], w2r[c][it][3], a3);
        }
        const float s = mt_wave_sum((a0 + a1) + (a2 + a3));
        if (lane == c) mine = s;
      }
      const int col = gw + lane * nw;
      if (lane < 2 && col < MT_D) mt_publish(gl + MG_X + col, epoch, (mine + Lw.b2[col]) + xbuf[col]);
    }
  }

  // ================= I: feats = LN_f(x); logits = E feats; masked arg-max of this workgroup's columns =================
  mt_fetch512(wr, p.emb, p.V, gw, nw, lane, 0);
  mt_gather(p.gran + (size_t)(MT_L - 1) * MG_LAYER + MG_X, MT_D, xbuf, epoch, err, t, lane);
  mt_layernorm(xbuf, p.lnf_g, p.lnf_b, ybuf, red, t, lane, wave);
  if (wg == 0) { p.feats[(size_t)it * MT_D + t] = ybuf[t]; p.feats[(size_t)it * MT_D + t + 256] = ybuf[t + 256]; }
  float bv = -INFINITY;
  int bi = 0x7fffffff;
  {
    const int nchunk = (p.V + 8 * nw - 1) / (8 * nw);
    for (int ch = 0; ch < nchunk; ++ch) {
      if (ch > 0) mt_fetch512(wr, p.emb, p.V, gw, nw, lane, ch * 8);
      float s = mt_dot512(wr, ybuf, lane);
      if (s != s) s = -INFINITY;                           // NaN -> -inf, still a candidate (as masked_argmax_kernel)
      const int col = gw + (ch * 8 + lane) * nw;
      if (lane < 8 && col < p.V && col != p.pad && !(ban_eos && col == p.eos)) {
        if (bi == 0x7fffffff || s > bv) { bv = s; bi = col; }      // this lane's columns ascend: first maximum wins
      }
    }
#pragma unroll
    for (int o = 4; o > 0; o >>= 1) {                      // candidates sit in lanes 0 .. 7
      const float ov = __shfl_xor(bv, o, 64);
      const int oi = __shfl_xor(bi, o, 64);
      if (oi != 0x7fffffff && (bi == 0x7fffffff || ov > bv || (ov == bv && oi < bi))) { bv = ov; bi = oi; }
    }
    if (lane == 0) { best_v[wave] = bv; best_i[wave] = bi; }
    __syncthreads();
    if (t == 0) {
      for (int w2 = 1; w2 < 4; ++w2) {
        const float ov = best_v[w2]; const int oi = best_i[w2];
        if (oi != 0x7fffffff && (bi == 0x7fffffff || ov > bv || (ov == bv && oi < bi))) { bv = ov; bi = oi; }
      }
      mt_publish(p.gran + MG_ARG + 2 * wg, epoch, bv);
      __hip_atomic_store(p.gran + MG_ARG + 2 * wg + 1, ((mt_u64)epoch << 32) | (unsigned)bi, MT_RLX);
    }
  }
  // ================= J: every workgroup picks the winner (workgroup 0 records it) =================
  {
    mt_gather(p.gran + MG_ARG, 2 * G, vec, epoch, err, t, lane);
    if (t == 0) {
      float fv = -INFINITY; int fi = 0x7fffffff;
      for (int g2 = 0; g2 < G; ++g2) {
        const float ov = vec[2 * g2]; const int oi = (int)__float_as_uint(vec[2 * g2 + 1]);
        if (oi != 0x7fffffff && (fi == 0x7fffffff || ov > fv || (ov == fv && oi < fi))) { fv = ov; fi = oi; }
      }
      // a bounded wait timed out somewhere in this launch (or an earlier one of this context): the token is not trustworthy.
      // -1 tells the host, which reads the chain back anyway, to redo the search with the launch-per-op form (model.hip).
      // The same when no column qualified (all logits NaN): the launch-per-op form then decides what such a step returns.
      const int nx = (__hip_atomic_load(err, MT_RLX) != 0u || fi == 0x7fffffff) ? -1 : (force_eos ? p.eos : fi);
      if (wg == 0) p.next[it] = nx;
      s_tok = nx;
    }
    __syncthreads();
    tk = s_tok;
    __syncthreads();
  }
  if (tk < 0 || tk == p.eos) break;              // the search is over (</s>), or broken (time-out): same decision in every workgroup
  }   // token loop
#endif
}

int launch_mt_step(const MtStepArgs& a, int G, hipStream_t stream) {
  if (G < 64 || G > MT_MAXG || (G & (G - 1)) != 0) return SS_ERR_ARG;   // >= 64: at most 8 columns (K = 512) / 2 (K = 2048) per wave
  hipLaunchKernelGGL(mt_step_kernel, dim3(G), dim3(256), 0, stream, a);
  SS_LAUNCH_CHECK();
  return SS_OK;
}

}  // namespace ss
