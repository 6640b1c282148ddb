// One Conformer layer of the INCREMENTAL streaming encoder (ss_encoder_stream_forward: the <= 48 non-final rows of a policy() call)
// as TWO persistent launches around the attention kernel instead of ten launches
// (reference: researches/chunk_unity/modules/conformer_layer.py:94-119, 152-164, 254-312; the reference recomputes the whole prefix per
// call -- agent/speech_to_speech.streamspeech.agent.py:425-435).
//
// Why: a read call of the 320-ms agent pushes ~8-24 rows through 12 layers -- 11 dependent launches per layer at 7-9 us each
// (tools/streaming_call_profile.py: 1.15 of the call's 1.6 ms), every one of them bound by launch + load latency, none by work
// (the fused batch FFN is SLOWER at these row counts: 38 vs 21 us).  Every op needs the WHOLE output of the previous one, so
// nothing fuses without a grid-wide exchange; as in mt_step.hip the exchange happens inside the launch:
//   launch A   phase 0  FFN1-a  all 64 workgroups: LN(x) -> LDS, H = SiLU(W1 slice . LN(x) + b1) for 32 hidden columns (in LDS),
//                               partial Y[n x 256] = H . W2 slice^T  -> part[w]                      (the hidden tensor never leaves the CU)
//              phase 1  FFN1-b  workgroup = row: x += 0.5 (sum_w part[w] + b2)   (64 partials added in workgroup order: fixed)
//              phase 2  QKV     48 workgroups, a 16-column tile each: LN(x) . Wqkv^T + b -> the layer's q|k|v cache rows
//   (attention_relpos kernel: its own launch -- K / V of ALL rows so far, key-split form)
//   launch B   phase 4  OUT     16 workgroups: x += Wo ctx + bo
//              phase 5  PW1     16 workgroups: GLU(Wpw1 LN(x)) -> the layer's GLU cache rows
//              phase 6  DW      workgroup = row: depthwise conv over the cached GLU rows (chunk-causal taps) + BatchNorm(eval) + SiLU
//              phase 7  PW2     16 workgroups: x += Wpw2 g
//              phase 8  FFN2-a, phase 9  FFN2-b + the layer's final LayerNorm
// Between phases: sc1 stores + a grid barrier on a monotone arrival counter (see "the exchange" below); every wait is bounded by TIME
// and counted in the scratch set's error word, nothing hangs.  All 64 workgroups must be resident -- the same condition as the persistent MT step,
// switched by the same setting (ss_mt_set_persistent); on a time-out the call is repeated with one launch per op.
// Arithmetic: exact-f32 MFMA; every K = 256 product is four 64-wide chains added in ascending order -- the CANON_KBLOCK summation of
// the packed path (gemm.hpp); W2's 2048 terms are 64 blocks of 32 added in order.
#include "enc_step.hpp"

#include <atomic>
#include <cstdio>

namespace ss {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

namespace {

constexpr int ES_XS = ES_D + 4;                 // LDS row stride of the staged rows (floats)
constexpr int ES_HS = 36;                       // LDS row stride of the 32-column hidden slice
constexpr int ES_SC1 = 16;                      // buffer cache policy: agent scope
constexpr unsigned long long ES_WAIT_TICKS = 20000000ull;   // 0.2 s of the 100-MHz wall clock: a workgroup of the launch is not resident

#define ES_RLX __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

// Tuning builds only (SS_EXTRA_FLAGS=-DES_TIMING=1): thread 0 of every workgroup stamps the 100-MHz wall clock at the kernel's entry, after
// every phase body and after every barrier into row 47 of its partial slot (unused: timing runs use n < 48); launch_enc_step then waits
// for the launch, reads the stamps back and prints per-event averages at exit.
#ifndef ES_TIMING
#define ES_TIMING 0
#endif
#if ES_TIMING
#define ES_STAMP(ev) do { if (t == 0) reinterpret_cast<unsigned long long*>(p.part + ((size_t)blockIdx.x * ES_MAXR + 47) * ES_D)[ev] = wall_clock64(); } while (0)
#else
#define ES_STAMP(ev) do { } while (0)
#endif

[[maybe_unused]] __device__ __forceinline__ f32x4 es_ld(const __amdgpu_buffer_rsrc_t rs, int byte_off) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, byte_off, 0, ES_SC1);
  return f32x4{__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3])};
}
[[maybe_unused]] __device__ __forceinline__ void es_st(const __amdgpu_buffer_rsrc_t rs, int byte_off, const f32x4 v) {
  const u32x4 u = {__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
  __builtin_amdgcn_raw_buffer_store_b128(u, rs, byte_off, 0, ES_SC1);
}
[[maybe_unused]] __device__ __forceinline__ float es_ld1(const __amdgpu_buffer_rsrc_t rs, int byte_off) {
  return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, byte_off, 0, ES_SC1));
}
[[maybe_unused]] __device__ __forceinline__ void es_st1(const __amdgpu_buffer_rsrc_t rs, int byte_off, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rs, byte_off, 0, ES_SC1);
}

// rows [0, n) of a [n][256] tensor (buffer rs, row stride ld floats) -> LDS xs[48][260], LayerNorm-ed when g != null (the arithmetic
// of rtlin.hip's staging: thread = row t / 4, 16-B chunks q, q + 4, ...); rows >= n read as zeros (buffer range check).
template <int NT>
[[maybe_unused]] __device__ __forceinline__ void es_stage(const __amdgpu_buffer_rsrc_t rs, int ld, const float* __restrict__ g, const float* __restrict__ b,
                                         float* xs, int t) {
  __syncthreads();                               // the previous phase's readers of xs are done
  if (t < 4 * 16 * NT) {
    const int row = t >> 2, q = t & 3;
    f32x4 v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = es_ld(rs, (row * ld + (i * 4 + q) * 4) * 4);
    if (g) {
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
      s += __shfl_xor(s, 1, 64);
      s += __shfl_xor(s, 2, 64);
      const float mean = s * (1.0f / ES_D);
      float qq = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d = v[i][e] - mean; qq += d * d; }
      qq += __shfl_xor(qq, 1, 64);
      qq += __shfl_xor(qq, 2, 64);
      const float rstd = 1.0f / sqrtf(qq * (1.0f / ES_D) + 1e-5f);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int c = i * 4 + q;
        const f32x4 gm = *reinterpret_cast<const f32x4*>(g + c * 4);
        const f32x4 bt = *reinterpret_cast<const f32x4*>(b + c * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[i][e] = (v[i][e] - mean) * rstd * gm[e] + bt[e];
      }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) *reinterpret_cast<f32x4*>(xs + row * ES_XS + ((i * 4 + q) << 2)) = v[i];
  }
  __syncthreads();
}

// Weight fragments of one 16-column tile over this wave's 64-wide k-block (K = 256 matrices: ldw = 256): four 16-B row segments per lane.
// Requested at the END of the previous phase (prefetch in the kernel), so that the L2 / HBM round trip runs under the barrier and the row staging.
[[maybe_unused]] __device__ __forceinline__ void es_wload(const float* __restrict__ W, int ldw, int n0, int k0, int r, int g, f32x4* wf) {
#pragma unroll
  for (int s = 0; s < 4; ++s) wf[s] = *reinterpret_cast<const f32x4*>(W + (size_t)(n0 + r) * ldw + k0 + 16 * s + 4 * g);
}

// acc[i] (row tile i = 0..NT-1) = W tile (fragments wf) . xs^T over this wave's 64-wide k-block
template <int NT>
[[maybe_unused]] __device__ __forceinline__ void es_gemm_block(const f32x4* wf, int k0, const float* xs, int r, int g, f32x4 (&acc)[NT]) {
#pragma unroll
  for (int i = 0; i < NT; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    f32x4 xa[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) xa[i] = *reinterpret_cast<const f32x4*>(xs + (16 * i + r) * ES_XS + k0 + 16 * s + 4 * g);
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int i = 0; i < NT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[s][e], xa[i][e], acc[i], 0, 0, 0);   // D = W . X^T
  }
}

// One 16-column tile of a K = 256 linear over the staged rows: the four waves take the four 64-wide k-blocks, the block sums meet in LDS
// and wave 0 adds them in ascending order (= CANON_KBLOCK); returns the tile in wave 0's registers (lane (r, g): row 16 i + r, columns 4 g ..)
template <int NT>
[[maybe_unused]] __device__ __forceinline__ void es_tile256(const f32x4* wf, const float* xs, f32x4* red, int wave, int lane, f32x4 (&out)[NT]) {
  const int r = lane & 15, g = lane >> 4;
  f32x4 acc[NT];
  es_gemm_block<NT>(wf, 64 * wave, xs, r, g, acc);
  __syncthreads();                               // (red may still be read by the previous tile's wave 0)
  if (wave > 0) {
#pragma unroll
    for (int i = 0; i < NT; ++i) red[((wave - 1) * 3 + i) * 64 + lane] = acc[i];
  }
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      f32x4 v = acc[i];
#pragma unroll
      for (int w2 = 0; w2 < 3; ++w2) {
        const f32x4 o = red[(w2 * 3 + i) * 64 + lane];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += o[e];
      }
      out[i] = v;
    }
  }
}

}  // namespace

// ---- the exchange -------------------------------------------------------------------------------------------------------------
// Data that another workgroup of the SAME launch needs goes out as sc1 (write-through) stores; a grid barrier follows every phase:
// s_waitcnt vmcnt(0) (the stores have left) -> workgroup barrier -> one relaxed agent-scope add on a monotone arrival counter, thread
// 0 spins until all 64 workgroups of the phase have arrived (bounded by TIME); consumers read with sc1 loads (per-XCD L2s are not
// coherent: MI355X_MICROARCH.md).  Measured on the 320-ms agent loop (tools/streaming_call_profile.py): 35 us per
// launch = 7.8 us per phase -- the price of four memory hops (store, drain, atomic, poll + load).  Three other forms were built and
// measured, all slower: {tag, value} granules as in mt_step.hip (every value its own flag: 54 us per launch -- 12 k 64-bit
// agent-scope loads per workgroup and phase cost more than the hops they save), one flag word per workgroup instead of the shared
// counter (51 us), phases as separate noinline functions (63 us); weight fragments requested before the wait bought nothing.
// The kernel is sensitive to register allocation: this arrangement compiles to 44 B of scratch per lane, the variants to 440-984 B.
[[maybe_unused]] __device__ __forceinline__ void es_arrive(const EsArgs& p, int t) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (t == 0) __hip_atomic_fetch_add(p.bar, 1u, ES_RLX);
}
[[maybe_unused]] __device__ __forceinline__ void es_wait(const EsArgs& p, int k, int t) {
  if (t == 0) {
    const unsigned target = p.bar_base + (unsigned)ES_G * (unsigned)(k + 1);
    const unsigned long long t0 = wall_clock64();
    while ((int)(__hip_atomic_load(p.bar, ES_RLX) - target) < 0) {
      if (__hip_atomic_load(p.err, ES_RLX) != 0u) break;
      if (wall_clock64() - t0 > ES_WAIT_TICKS) {
        atomicAdd(p.err, 1u);
        __hip_atomic_fetch_add(p.err_host, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        break;
      }
      __builtin_amdgcn_s_sleep(2);
    }
  }
  __syncthreads();
}

template <int NT>
__global__ __launch_bounds__(256, 1) void enc_step_kernel(const EsArgs p) {
#if __HIP_DEVICE_COMPILE__
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* xs = smem;                                              // [48][260]
  f32x4* red = reinterpret_cast<f32x4*>(smem + ES_MAXR * ES_XS);  // [3 waves][3 tiles][64 lanes] x 2 column tiles
  float* hs = smem + ES_MAXR * ES_XS + 2 * 9 * 64 * 4;           // [48][36]
  float* rowred = hs + ES_MAXR * ES_HS;                          // [16]
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int r = lane & 15, g = lane >> 4;
  const int w = blockIdx.x, n = p.n;
  const EsLayerW& L = p.w;
  const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, n * ES_D * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsP = __builtin_amdgcn_make_buffer_rsrc((void*)p.part, 0, ES_G * ES_MAXR * ES_D * 4, 0x00020000);
  int nbar = 0;

  // The weight fragments of a phase are requested BEFORE the wait of the barrier that precedes it: every layer's weights are cold, and as
  // loads behind the row staging their L2 / HBM round trips were most of a phase body (in-kernel stamps, -DES_TIMING=1: the FFN-a body 7 us
  // with six dependent weight round trips, 3.6-5 with the requests ahead of the wait).  16 x 16 B per lane, consumed by the phase body.
  f32x4 pf[16];
  auto prefetch = [&](int ph) {
    switch (ph) {
      case 0: case 8: {
        const float* W1 = ph == 0 ? L.ffn1_w1 : L.ffn2_w1;
        const float* W2 = ph == 0 ? L.ffn1_w2 : L.ffn2_w2;
        es_wload(W1, ES_D, 32 * w, 64 * wave, r, g, pf);
        es_wload(W1, ES_D, 32 * w + 16, 64 * wave, r, g, pf + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int s = 0; s < 2; ++s)
            pf[8 + 2 * j + s] = *reinterpret_cast<const f32x4*>(W2 + (size_t)((4 * wave + j) * 16 + r) * ES_F + 32 * w + 16 * s + 4 * g);
      } break;
      case 2: if (w < 3 * ES_D / 16) es_wload(L.qkv_w, ES_D, 16 * w, 64 * wave, r, g, pf); break;
      case 4: if (w < ES_D / 16) es_wload(L.out_w, ES_D, 16 * w, 64 * wave, r, g, pf); break;
      case 5:
        if (w < ES_D / 16) { es_wload(L.pw1_w, ES_D, 32 * w, 64 * wave, r, g, pf); es_wload(L.pw1_w, ES_D, 32 * w + 16, 64 * wave, r, g, pf + 4); }
        break;
      case 7: if (w < ES_D / 16) es_wload(L.pw2_w, ES_D, 16 * w, 64 * wave, r, g, pf); break;
      default: break;
    }
    asm volatile("" ::: "memory");                 // the requests stay where they are written: ahead of the wait
  };

  // FFN halves: `a` on every workgroup, `b` on workgroup = row
  auto ffn_a = [&](const float* lg, const float* lb, const float* b1) {
    es_stage<NT>(rsX, ES_D, lg, lb, xs, t);
    // H = SiLU(W1[32 w .. 32 w + 31] . LN(x)^T + b1): two column tiles, each wave one k-block of each (fragments pf[0..7])
    f32x4 a0[NT], a1[NT];
    es_gemm_block<NT>(pf, 64 * wave, xs, r, g, a0);
    es_gemm_block<NT>(pf + 4, 64 * wave, xs, r, g, a1);
    if (wave > 0) {
#pragma unroll
      for (int i = 0; i < NT; ++i) { red[((wave - 1) * 3 + i) * 64 + lane] = a0[i]; red[(9 + (wave - 1) * 3 + i) * 64 + lane] = a1[i]; }
    }
    __syncthreads();
    if (wave == 0) {
      const f32x4 bb0 = *reinterpret_cast<const f32x4*>(b1 + 32 * w + 4 * g), bb1 = *reinterpret_cast<const f32x4*>(b1 + 32 * w + 16 + 4 * g);
#pragma unroll
      for (int i = 0; i < NT; ++i) {
        f32x4 v0 = a0[i], v1 = a1[i];
#pragma unroll
        for (int w2 = 0; w2 < 3; ++w2) {
          const f32x4 o0 = red[(w2 * 3 + i) * 64 + lane], o1 = red[(9 + w2 * 3 + i) * 64 + lane];
#pragma unroll
          for (int e = 0; e < 4; ++e) { v0[e] += o0[e]; v1[e] += o1[e]; }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float u0 = v0[e] + bb0[e], u1 = v1[e] + bb1[e];
          v0[e] = u0 / (1.0f + expf(-u0));
          v1[e] = u1 / (1.0f + expf(-u1));
        }
        *reinterpret_cast<f32x4*>(hs + (16 * i + r) * ES_HS + 4 * g) = v0;
        *reinterpret_cast<f32x4*>(hs + (16 * i + r) * ES_HS + 16 + 4 * g) = v1;
      }
    }
    __syncthreads();
    // partial Y[48 x 256] = H[48 x 32] . W2[:, 32 w .. 32 w + 31]^T: wave takes column tiles 4 wave .. 4 wave + 3 (fragments pf[8..15])
    f32x4 ha[NT][2];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
      for (int s = 0; s < 2; ++s) ha[i][s] = *reinterpret_cast<const f32x4*>(hs + (16 * i + r) * ES_HS + 16 * s + 4 * g);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int ct = 4 * wave + j;
      const f32x4* wf = pf + 8 + 2 * j;
      f32x4 acc[NT];
#pragma unroll
      for (int i = 0; i < NT; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int i = 0; i < NT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[s][e], ha[i][s][e], acc[i], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < NT; ++i)
        if (16 * i + r < n) es_st(rsP, ((w * ES_MAXR + 16 * i + r) * ES_D + ct * 16 + 4 * g) * 4, acc[i]);
    }
  };
  auto ffn_b = [&](const float* b2, const float* ln_g, const float* ln_b) {
    if (w >= n) return;
    float part[ES_G];
#pragma unroll
    for (int j = 0; j < ES_G; ++j) part[j] = es_ld1(rsP, ((j * ES_MAXR + w) * ES_D + t) * 4);
    float s = part[0];
#pragma unroll
    for (int j = 1; j < ES_G; ++j) s += part[j];
    float v = (s + b2[t]) * 0.5f + es_ld1(rsX, (w * ES_D + t) * 4);
    if (ln_g) {                                                  // the layer's final LayerNorm over the row (two-pass)
      float a = v;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
      if (lane == 0) rowred[wave] = a;
      __syncthreads();
      const float mean = ((rowred[0] + rowred[1]) + (rowred[2] + rowred[3])) * (1.0f / ES_D);
      const float d = v - mean;
      float q = d * d;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
      if (lane == 0) rowred[4 + wave] = q;
      __syncthreads();
      const float rstd = 1.0f / sqrtf(((rowred[4] + rowred[5]) + (rowred[6] + rowred[7])) * (1.0f / ES_D) + 1e-5f);
      v = d * rstd * ln_g[t] + ln_b[t];
    }
    es_st1(rsX, (w * ES_D + t) * 4, v);
  };
  // x += W tile . A^T + bias for one 16-column tile (attention output / pointwise conv 2): A = `src` rows, no LayerNorm
  auto proj_residual = [&](const __amdgpu_buffer_rsrc_t rsA, const float* bias) {
    if (w >= ES_D / 16) return;
    es_stage<NT>(rsA, ES_D, nullptr, nullptr, xs, t);
    f32x4 out[NT];
    es_tile256<NT>(pf, xs, red, wave, lane, out);
    if (wave == 0) {
      f32x4 bb = f32x4{0.f, 0.f, 0.f, 0.f};
      if (bias) bb = *reinterpret_cast<const f32x4*>(bias + 16 * w + 4 * g);
#pragma unroll
      for (int i = 0; i < NT; ++i) {
        const int row = 16 * i + r;
        if (row < n) {
          const f32x4 xr = es_ld(rsX, (row * ES_D + 16 * w + 4 * g) * 4);
          f32x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = (out[i][e] + bb[e]) + xr[e];
          es_st(rsX, (row * ES_D + 16 * w + 4 * g) * 4, v);
        }
      }
    }
  };

  [[maybe_unused]] int ev = 0;
  ES_STAMP(ev++);
  prefetch(p.ph0);
  for (int ph = p.ph0; ph <= p.ph1; ++ph) {
    if (ph > p.ph0) {
      ES_STAMP(ev++);
      es_arrive(p, t);
      prefetch(ph);
      es_wait(p, nbar++, t);
      ES_STAMP(ev++);
    }
    switch (ph) {
      case 0: ffn_a(L.ffn1_ln_g, L.ffn1_ln_b, L.ffn1_b1); break;
      case 1: ffn_b(L.ffn1_b2, nullptr, nullptr); break;
      case 2: {                                                  // q | k | v rows of the layer cache
        if (w >= 3 * ES_D / 16) break;
        es_stage<NT>(rsX, ES_D, L.attn_ln_g, L.attn_ln_b, xs, t);
        f32x4 out[NT];
        es_tile256<NT>(pf, xs, red, wave, lane, out);
        if (wave == 0) {
          const f32x4 bb = *reinterpret_cast<const f32x4*>(L.qkv_b + 16 * w + 4 * g);
          const __amdgpu_buffer_rsrc_t rsQ = __builtin_amdgcn_make_buffer_rsrc((void*)(p.qkv + (size_t)p.r0 * 3 * ES_D), 0, n * 3 * ES_D * 4, 0x00020000);
#pragma unroll
          for (int i = 0; i < NT; ++i) {
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = out[i][e] + bb[e];
            if (16 * i + r < n) es_st(rsQ, ((16 * i + r) * 3 * ES_D + 16 * w + 4 * g) * 4, v);
          }
        }
      } break;
      case 4: {
        const __amdgpu_buffer_rsrc_t rsH = __builtin_amdgcn_make_buffer_rsrc((void*)p.hctx, 0, n * ES_D * 4, 0x00020000);
        proj_residual(rsH, L.out_b);
      } break;
      case 5: {                                                  // GLU(Wpw1 LN(x)): weight rows [32 w, 32 w + 16) value | [.. + 16, .. + 32) gate
        if (w >= ES_D / 16) break;
        es_stage<NT>(rsX, ES_D, L.conv_ln_g, L.conv_ln_b, xs, t);
        f32x4 val[NT], gate[NT];
        es_tile256<NT>(pf, xs, red, wave, lane, val);
        es_tile256<NT>(pf + 4, xs, red, wave, lane, gate);
        if (wave == 0) {
          const __amdgpu_buffer_rsrc_t rsG = __builtin_amdgcn_make_buffer_rsrc((void*)(p.glu + (size_t)p.r0 * ES_D), 0, n * ES_D * 4, 0x00020000);
#pragma unroll
          for (int i = 0; i < NT; ++i) {
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = val[i][e] * (1.0f / (1.0f + expf(-gate[i][e])));
            if (16 * i + r < n) es_st(rsG, ((16 * i + r) * ES_D + 16 * w + 4 * g) * 4, v);
          }
        }
      } break;
      case 6: {                                                  // depthwise conv + BatchNorm(eval) + SiLU of row r0 + w, channel t
        if (w >= n) break;
        const int tt = p.r0 + w, half = p.dwk / 2;
        int lim = p.T2;
        if (p.cchunk > 0) { const int ce = (tt / p.cchunk + 1) * p.cchunk; if (ce < lim) lim = ce; }
        const int jmax = min(p.dwk, lim - (tt - half));
        const __amdgpu_buffer_rsrc_t rsG = __builtin_amdgcn_make_buffer_rsrc((void*)p.glu, 0, p.T2 * ES_D * 4, 0x00020000);
        float sv[31];
#pragma unroll
        for (int j = 0; j < 31; ++j) {
          const int tin = tt - half + j;
          sv[j] = (j < p.dwk && tin >= 0) ? es_ld1(rsG, (tin * ES_D + t) * 4) : 0.f;     // rows >= T2: out of range -> 0 (and not visible anyway)
        }
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < 31; ++j) {
          const float wj = j < p.dwk ? L.dw_wt[j * ES_D + t] : 0.f;
          const float a2 = fmaf(wj, sv[j], acc);
          acc = (j < jmax) ? a2 : acc;
        }
        const float v = (acc - L.bn_mean[t]) * (1.0f / sqrtf(L.bn_var[t] + 1e-5f)) * L.bn_g[t] + L.bn_b[t];
        const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc((void*)p.g2, 0, n * ES_D * 4, 0x00020000);
        es_st1(rsO, (w * ES_D + t) * 4, v / (1.0f + expf(-v)));
      } break;
      case 7: {
        const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc((void*)p.g2, 0, n * ES_D * 4, 0x00020000);
        proj_residual(rsO, nullptr);
      } break;
      case 8: ffn_a(L.ffn2_ln_g, L.ffn2_ln_b, L.ffn2_b1); break;
      case 9: ffn_b(L.ffn2_b2, L.final_ln_g, L.final_ln_b); break;
      default: break;
    }
  }
  ES_STAMP(ev++);
#endif
}

size_t enc_step_scratch_bytes() { return (size_t)(ES_G * ES_MAXR * ES_D + ES_MAXR * ES_D) * sizeof(float) + 512; }   // partials + g2 + counter + error word
size_t enc_step_lds_bytes() { return (size_t)(ES_MAXR * ES_XS + 2 * 9 * 64 * 4 + ES_MAXR * ES_HS + 16) * sizeof(float); }

static std::atomic<long long> g_enc_step_launches{0};
long long enc_step_launch_count() { return g_enc_step_launches.load(std::memory_order_relaxed); }

#if ES_TIMING
namespace {
struct EsTiming {
  double sum[2][16][3] = {};      // [launch A | B][event][first arrival, mean, last arrival] in us after the first workgroup's entry
  long long launches[2] = {};
  ~EsTiming() {
    for (int k = 0; k < 2; ++k) {
      if (!launches[k]) continue;
      const int nev = k == 0 ? 6 : 12;
      fprintf(stderr, "enc_step timing, launch %c (%lld launches; us after the first workgroup's entry: first / mean / last workgroup)\n", k ? 'B' : 'A', launches[k]);
      for (int e = 0; e < nev; ++e) {
        const char* what = e == 0 ? "entry" : e == nev - 1 ? "exit" : (e & 1) ? "phase body done" : "barrier passed";
        fprintf(stderr, "  event %2d %-16s %7.2f %7.2f %7.2f\n", e, what, sum[k][e][0] / launches[k], sum[k][e][1] / launches[k], sum[k][e][2] / launches[k]);
      }
    }
  }
  void add(const EsArgs& a, hipStream_t stream) {
    static long long seen = 0;
    hipStreamSynchronize(stream);
    if (++seen <= 200) return;                                   // warm-up
    const int k = a.ph0 == 0 ? 0 : 1, nev = k == 0 ? 6 : 12;
    static unsigned long long st[ES_G][16];
    for (int w = 0; w < ES_G; ++w) hipMemcpy(st[w], a.part + ((size_t)w * ES_MAXR + 47) * ES_D, 16 * 8, hipMemcpyDeviceToHost);
    unsigned long long t0 = ~0ull;
    for (int w = 0; w < ES_G; ++w) t0 = st[w][0] < t0 ? st[w][0] : t0;
    for (int e = 0; e < nev; ++e) {
      double lo = 1e30, hi = -1e30, m = 0;
      for (int w = 0; w < ES_G; ++w) { const double v = (double)(st[w][e] - t0) * 0.01; lo = v < lo ? v : lo; hi = v > hi ? v : hi; m += v; }
      sum[k][e][0] += lo; sum[k][e][1] += m / ES_G; sum[k][e][2] += hi;
    }
    ++launches[k];
  }
};
EsTiming g_es_timing;
}  // namespace
#endif

template <int NT>
static int launch_enc_step_nt(const EsArgs& a, hipStream_t stream) {
  const size_t lds = enc_step_lds_bytes();
  SS_MAX_LDS_ONCE((&enc_step_kernel<NT>), lds);
  hipLaunchKernelGGL(enc_step_kernel<NT>, dim3(ES_G), dim3(256), lds, stream, a);
  SS_LAUNCH_CHECK();
#if ES_TIMING
  g_es_timing.add(a, stream);
#endif
  return SS_OK;
}

// The kernel is compiled for 1, 2 and 3 MFMA row tiles: a read call of the 320-ms agent recomputes ~8-16 rows, and the 48-row form spent
// two thirds of its MFMAs (and of its staging loads) on rows that do not exist.  Per row the arithmetic is the same in all three.
int launch_enc_step(const EsArgs& a, hipStream_t stream) {
  if (a.n <= 0 || a.n > ES_MAXR || a.ph0 > a.ph1 || a.dwk > 31 || !a.err || !a.err_host || !a.bar || !a.part || !a.g2) return SS_ERR_ARG;
  const int nt = (a.n + 15) / 16;
  const int rc = nt == 1 ? launch_enc_step_nt<1>(a, stream) : nt == 2 ? launch_enc_step_nt<2>(a, stream) : launch_enc_step_nt<3>(a, stream);
  if (rc != SS_OK) return rc;
  g_enc_step_launches.fetch_add(1, std::memory_order_relaxed);
  return SS_OK;
}

}  // namespace ss
