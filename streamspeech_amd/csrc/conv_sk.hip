// Persistent stream-K implicit-GEMM Conv1d for gfx950 (exact-f32 MFMA), the big-problem path of
// launch_conv_gemm: HiFi-GAN resblock / upsampling convs over a packed batch of utterances
// (reference fairseq/models/text_to_speech/hifigan.py:52-172).
//
// Why a second kernel: at batch scale the 32x64-tile kernel is bound by everything *except* the
// matrix cores (24 B/cycle/CU of L2->LDS traffic, LDS at 56 %, one barrier per 512 MFMA cycles), and
// 128x128 tiles -- which need 3x less of all of that -- leave half the chip idle because a launch
// has only ~300 of them for 512 resident workgroups (profiles/r01_tile_sweep_batch.txt).  Here:
//   * 128 x BN tiles (BN = 128 | 64), BK = 32, 4 waves as 2x2, each wave 64 x BN/2 (TM=4, TN=BN/32).
//   * stream-K: the linear (tile, k-step) space is cut into G equal contiguous ranges, one per
//     workgroup (G = 2 per CU, all resident).  A workgroup that owns a tile's last k-step is its
//     finisher; earlier owners park their partial tile in a workspace slot and raise a flag
//     (agent-scope release); the finisher adds the partials in workgroup order (fixed order ->
//     deterministic) and runs the epilogue.  Logical workgroup ids come from an atomic ticket, so a
//     finisher only ever waits on workgroups that are already running (no dispatch-order assumption).
//   * global -> LDS with global_load_lds_dwordx4 (no staging registers, no ds_write pass).  The LDS
//     image is lane-linear (HW rule), so the bank swizzle is applied to the per-lane SOURCE address
//     and again at the ds_read_b128: 16-B chunk c of row r sits at chunk position c ^ ((r>>1)&7).
//     Out-of-segment rows (conv zero padding, ragged batch edges) read a zero page instead.
//   * one barrier per k-step: wait own loads -> barrier -> issue next step's loads -> 128 MFMAs.
//   * the leaky-ReLU on the conv input is applied to the A fragments after the ds_read
//     (max(v, slope*v), 8 VALU per 64 MFMAs).
#include "gemm.hpp"

#include <map>
#include <mutex>

namespace ss {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
[[maybe_unused]] constexpr int SK_SC1 = 16;            // buffer cache policy: sc1 = agent scope (writes through / reads past the per-XCD L2)
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef __attribute__((address_space(1))) const void* gbl_ptr_t;

struct SkArgs {
  float* ws;            // [G][128*BN] partial tiles
  unsigned* sync;       // [0..7] ticket counter of XCD group x, [8] time-out counter, [16 + w] flag of logical workgroup w
  unsigned base[8];     // ticket value of rank 0 of each group for this launch
  unsigned epoch;       // flag value meaning "partial of this launch is in place"
  int G;                // workgroups (multiple of NG)
  int NG;               // 8: tiles are split over the 8 XCDs first (workgroup i runs on XCD i % 8), 1: no grouping
};

constexpr int SK_BM = 128, SK_BK = 32, SK_MAXG = 512;
[[maybe_unused]] constexpr int SK_FLAG0 = 16;
[[maybe_unused]] constexpr unsigned SK_SPIN_LIMIT = 1u << 22;
[[maybe_unused]] constexpr int SK_NUM_RECORDS = 0x7ffffff0;          // buffer range: every real offset is below, SK_OOB is above
[[maybe_unused]] constexpr unsigned SK_OOB = 0x80000000u;

__device__ __attribute__((noinline)) float sk_act_slow(float v, int act) {
  if (act == ACT_SILU) return v / (1.0f + expf(-v));
  if (act == ACT_TANH) return tanhf(v);
  return v;
}

template <int BN, bool LRELU>
__global__ __launch_bounds__(256, 2) void conv_sk_kernel(const GemmArgs p, const SkArgs q) {
#if __HIP_DEVICE_COMPILE__   // the buffer-resource builtins have no host-pass meaning (the stub would not be emitted)
  constexpr int BM = SK_BM, BK = SK_BK;
  constexpr int TM = 4, TN = BN / 32;          // 16x16 MFMA tiles per wave (wave tile 64 x BN/2)
  constexpr int NWI = BN / 32;                 // W glds instructions per wave per k-step (A: 4)
  constexpr int STAGE = (BM + BN) * BK;        // floats per LDS stage
  extern __shared__ __attribute__((aligned(16))) float smem[];
  int* s_lo = reinterpret_cast<int*>(smem + 2 * STAGE);
  int* s_hi = s_lo + BM;
  int* s_misc = s_hi + BM;

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int r = lane & 15, g = lane >> 4;

  // XCD grouping: the hardware deals workgroup i to XCD i % 8, each XCD has its own L2.  Whole tiles
  // are first split over 8 groups and stream-K runs inside a group, so the N tiles of an M tile and
  // the neighbouring M tiles (shared conv halo rows, the tile's A rows re-read per tap) hit one L2,
  // and no fix-up crosses XCDs.  The grouping only steers locality: correctness needs nothing from
  // the hardware mapping (each group has its own ticket counter and exactly G/NG members).
  const int grp = q.NG > 1 ? (int)(blockIdx.x & 7) : 0;
  const int Gg = q.G / q.NG;
  if (t == 0) s_misc[0] = (int)(atomicAdd(q.sync + grp, 1u) - q.base[grp]);
  __syncthreads();
  const int rank = __builtin_amdgcn_readfirstlane(s_misc[0]);
  // a ticket outside [0, Gg) means the host's ticket base and the device counter disagree (two host threads driving one
  // context): count it like a bounded-wait time-out and leave, instead of indexing tiles / workspace slots with it
  if ((unsigned)rank >= (unsigned)Gg) { if (t == 0) atomicAdd(q.sync + 8, 1u); return; }
  const int w = grp * Gg + rank;                 // logical id: workspace slot / flag index

  const int kpt = p.Cin / BK;
  const int nk = p.taps * kpt;
  const int Ktot = p.taps * p.Cin;
  const int tiles_n = p.N / BN;
  const int tiles_m = (p.M + BM - 1) / BM;
  const long long tiles = (long long)tiles_m * tiles_n;
  const long long t_lo = tiles * grp / q.NG, t_hi = tiles * (grp + 1) / q.NG;
  const long long ubase = t_lo * nk;               // first unit of the group
  const long long U = (t_hi - t_lo) * nk;          // units of the group, split over its Gg workgroups
  const long long u0 = ubase + (long long)rank * U / Gg;
  long long ue = ubase + (long long)(rank + 1) * U / Gg;     // the range is walked from its END: see below

  const float slope = p.in_slope;
  // Buffer resources for the LDS-DMA loads: per-lane byte offsets stay constant over a tile and the
  // k-step (tap shift, channel block) goes into the wave-uniform soffset, so staging costs 4 VALU
  // per A row and none per W row.  Rows outside their utterance get an offset beyond num_records:
  // the buffer range check then returns zeros (conv zero padding).  The A base is moved back by
  // `pad` rows so that every valid offset is non-negative.
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(reinterpret_cast<uintptr_t>(p.A) - (uintptr_t)p.pad * p.lda * sizeof(float)), 0, SK_NUM_RECORDS, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, SK_NUM_RECORDS, 0x00020000);

  // per-lane LDS read offsets (floats) inside a stage: row (r) * 32 + swizzled chunk
  const int swz = (r >> 1) & 7;
  const int rdA0 = (wm * 64 + r) * BK + ((g ^ swz) << 2);
  const int rdA1 = (wm * 64 + r) * BK + (((4 + g) ^ swz) << 2);
  const int rdW0 = BM * BK + (wn * (BN / 2) + r) * BK + ((g ^ swz) << 2);
  const int rdW1 = BM * BK + (wn * (BN / 2) + r) * BK + (((4 + g) ^ swz) << 2);
  // per-lane staging roles: instruction j of this wave covers 8 rows x 8 chunks
  const int st_row = lane >> 3, st_pos = lane & 7;

  int cur_tm = -1;
  // Tiles of the range are processed last-to-first.  The only tile a workgroup can leave unfinished
  // is the last one of its range (it owns the head k-steps, a later workgroup owns the end), so its
  // partial is parked before anything else; the tile it must finish is the first one of its range
  // and comes last -- by then the earlier workgroups (lower tickets, already running) have parked
  // theirs.  Nobody ever waits on a workgroup that is itself waiting: no dependency chains.
  while (ue > u0) {
    const int tile = (int)((ue - 1) / nk);
    const long long ut0 = (long long)tile * nk;
    const int ka = (int)(max(u0, ut0) - ut0);
    const int kb = (int)(ue - ut0);
    ue = ut0 + ka;
    const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;

    __syncthreads();                     // previous tile's LDS reads (stages, row bounds) are done
    if (tm != cur_tm) {
      cur_tm = tm;
      if (t < BM) {
        const int m = m0 + t;
        int lo = 0, hi = 0;
        if (m < p.M) {
          if (p.nseg > 0) {
            // segments are contiguous and ascending (same_rows): binary search for the one that holds row m
            int a = 0, bsz = p.nseg;
            while (bsz > 1) {
              const int half = bsz >> 1;
              if (p.segs[4 * (a + half)] <= m) a += half;
              bsz -= half;
            }
            const int st = p.segs[4 * a], ln = p.segs[4 * a + 1];
            if (m >= st && m < st + ln) { lo = st; hi = st + ln; }
          } else {
            hi = p.in_len;
          }
        }
        s_lo[t] = lo; s_hi[t] = hi;
      }
      __syncthreads();
    }
    int a_rin0[4], a_lo[4], a_hi[4];
    unsigned a_off[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = wave * 32 + j * 8 + st_row;
      a_rin0[j] = m0 + row - p.pad;
      a_lo[j] = s_lo[row]; a_hi[j] = s_hi[row];
      a_off[j] = (unsigned)(((m0 + row) * p.lda + ((st_pos ^ ((row >> 1) & 7)) << 2)) * 4);
    }
    unsigned w_off[NWI];
#pragma unroll
    for (int j = 0; j < NWI; ++j) {
      const int nrow = wave * (BN / 4) + j * 8 + st_row;
      w_off[j] = (unsigned)(((n0 + nrow) * Ktot + ((st_pos ^ ((nrow >> 1) & 7)) << 2)) * 4);
    }
    // k-step s = (channel block s / taps, tap s % taps): the taps of one 32-channel block run back to
    // back, so the (128 + halo) x 128-B slab of A rows they share is re-read from L1/L2 while it is
    // still there (tap-major order put 2 MB of other traffic per XCD between two reads of a row).
    int nci = (ka / p.taps) * BK, ntap = ka - (ka / p.taps) * p.taps;     // of the next step to stage
    auto issue = [&](int kstep, int stage) {
      float* sA = smem + stage * STAGE + (wave * 32) * BK;
      float* sW = smem + stage * STAGE + BM * BK + (wave * (BN / 4)) * BK;
      const int shift = ntap * p.dil;
      const int soffA = (shift * p.lda + nci) * 4;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int rin = a_rin0[j] + shift;
        const bool ok = rin >= a_lo[j] && rin < a_hi[j];
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr_t)(sA + j * 8 * BK), 16, ok ? a_off[j] : SK_OOB, soffA, 0, 0);
      }
      const int soffW = (ntap * p.Cin + nci) * 4;
#pragma unroll
      for (int j = 0; j < NWI; ++j)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lds_ptr_t)(sW + j * 8 * BK), 16, w_off[j], soffW, 0, 0);
      if (++ntap >= p.taps) { ntap = 0; nci += BK; }
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    issue(ka, 0);
    for (int k = ka; k < kb; ++k) {
      const int cur = (k - ka) & 1;
      {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();                   // step k landed for everyone; everyone finished reading stage cur^1
      }
      const float* S = smem + cur * STAGE;
#ifndef SK_VARIANT
#define SK_VARIANT 6   // bit 0: s_setprio around the MFMA blocks (neutral); bit 1: first-half fragments before the next DMA issue (+2 %); bit 2: second-half fragments prefetched under the first half's MFMAs (+3 % at batch 32)
#endif
      auto load_frags = [&](int kk, f32x4 (&af)[TM], f32x4 (&bf)[TN]) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          af[i] = *reinterpret_cast<const f32x4*>(S + (kk ? rdA1 : rdA0) + i * 16 * BK);
          if (LRELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) af[i][e] = fmaxf(af[i][e], af[i][e] * slope);
          }
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const f32x4*>(S + (kk ? rdW1 : rdW0) + j * 16 * BK);
      };
      auto mma = [&](f32x4 (&af)[TM], f32x4 (&bf)[TN]) {
#if SK_VARIANT & 1
        __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[j][e], af[i][e], acc[i][j], 0, 0, 0);   // D = W.A^T: see epilogue
#if SK_VARIANT & 1
        __builtin_amdgcn_s_setprio(0);
#endif
      };
#if SK_VARIANT & 2
      {   // fragments of the first half first (shortest path from the barrier to the first MFMA), then the next step's DMA
        f32x4 af[TM], bf[TN];
        load_frags(0, af, bf);
        __builtin_amdgcn_sched_barrier(0);
        if (k + 1 < kb) issue(k + 1, cur ^ 1);
        __builtin_amdgcn_sched_barrier(0);
        {
#if SK_VARIANT & 4
          f32x4 af1[TM], bf1[TN];
          load_frags(1, af1, bf1);            // second-half fragments in flight under the first half's MFMAs
          __builtin_amdgcn_sched_barrier(0);
          mma(af, bf);
          mma(af1, bf1);
#else
          mma(af, bf);
          f32x4 af1[TM], bf1[TN];
          load_frags(1, af1, bf1);
          mma(af1, bf1);
#endif
        }
      }
#else
      if (k + 1 < kb) issue(k + 1, cur ^ 1);
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        f32x4 af[TM], bf[TN];
        load_frags(kk, af, bf);
        mma(af, bf);
      }
#endif
    }

    bool has_end = kb == nk;
    if (!has_end) {
      // ---- contributor: park the partial tile, raise the flag ----
      // Partials move as sc1 (agent-scope) b128 buffer stores / loads, flags as sc1 relaxed atomics: they write through / read
      // past the per-XCD L2, so no L2 write-back or invalidate (which would evict the weights every
      // other workgroup of the XCD is streaming) is needed.  Order: stores complete (vmcnt 0) ->
      // workgroup barrier -> flag.
      const __amdgpu_buffer_rsrc_t rsP = __builtin_amdgcn_make_buffer_rsrc((void*)(q.ws + (size_t)w * (BM * BN)), 0, BM * BN * 4, 0x00020000);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          u32x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = __float_as_uint(acc[i][j][e]);
          __builtin_amdgcn_raw_buffer_store_b128(v, rsP, (((wave * TM + i) * TN + j) * 64 + lane) * 16, 0, SK_SC1);
        }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (t == 0) __hip_atomic_store(q.sync + SK_FLAG0 + w, q.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      continue;
    }
    if (ka > 0) {
      // ---- finisher: collect the partials of the workgroups that own k-steps [0, ka) of this tile ----
      const int wf = grp * Gg + (int)(((ut0 - ubase + 1) * Gg - 1) / U);   // workgroup (of this group) that owns the tile's first unit
      if (t == 0) {
        for (int ww = wf; ww < w; ++ww) {
          unsigned spins = 0;
          while (__hip_atomic_load(q.sync + SK_FLAG0 + ww, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != q.epoch) {
            __builtin_amdgcn_s_sleep(8);
            if (++spins > SK_SPIN_LIMIT) { atomicAdd(q.sync + 8, 1u); break; }
          }
        }
      }
      __syncthreads();
      for (int ww = wf; ww < w; ++ww) {      // fixed order: ((mine + P[wf]) + P[wf+1]) + ...
        const __amdgpu_buffer_rsrc_t rsP = __builtin_amdgcn_make_buffer_rsrc((void*)(q.ws + (size_t)ww * (BM * BN)), 0, BM * BN * 4, 0x00020000);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          u32x4 o[TN];
#pragma unroll
          for (int j = 0; j < TN; ++j)
            o[j] = __builtin_amdgcn_raw_buffer_load_b128(rsP, (((wave * TM + i) * TN + j) * 64 + lane) * 16, 0, SK_SC1);
#pragma unroll
          for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][j][e] += __uint_as_float(o[j][e]);
          __builtin_amdgcn_sched_barrier(0);   // TN loads in flight per step: keeps the register budget of the main loop
        }
      }
    }

    // ---- epilogue (same operation order as conv_gemm_kernel) ----
    // The MFMAs were issued with the operands swapped (D = W_tile . A_tile^T), so in the C/D layout
    // (col = lane&15, row = 4*(lane>>4) + reg) a lane holds 4 CONSECUTIVE output channels of ONE row:
    // bias / residual loads and the stores are float4 (4x fewer memory instructions than the scalar
    // column-per-lane form, same 64-B segments).  Needs ldc/ldr/ldr2/ldc2 % 4 == 0 (checked on the host).
    // Every option is ONE wave-uniform branch around a loop over a row tile's TN x 4 elements, and SiLU / tanh are out
    // of line: a per-element switch with expf / tanhf / division inlined made this a 57-KB kernel whose epilogue
    // thrashed the instruction cache (profiles/r02_sk2_ablation_epilogue.txt).
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int m = m0 + wm * 64 + i * 16 + r;
      if (m >= p.M) continue;
      f32x4 v[TN];
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        v[j] = acc[i][j];
        if (p.bias) {
          const f32x4 b = *reinterpret_cast<const f32x4*>(p.bias + n0 + wn * (BN / 2) + j * 16 + g * 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[j][e] += b[e];
        }
      }
      if (p.act == ACT_LRELU) {
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) v[j][e] = v[j][e] > 0.f ? v[j][e] : v[j][e] * p.act_slope;
      } else if (p.act == ACT_RELU) {
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) v[j][e] = fmaxf(v[j][e], 0.f);
      } else if (p.act != ACT_NONE) {
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) v[j][e] = sk_act_slow(v[j][e], p.act);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) v[j][e] *= p.alpha;
      if (p.R) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const f32x4 rr = *reinterpret_cast<const f32x4*>(p.R + (size_t)m * p.ldr + n0 + wn * (BN / 2) + j * 16 + g * 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[j][e] += rr[e];
        }
      }
      if (p.R2) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const f32x4 rr = *reinterpret_cast<const f32x4*>(p.R2 + (size_t)m * p.ldr2 + n0 + wn * (BN / 2) + j * 16 + g * 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[j][e] = rr[e] + v[j][e];
        }
      }
      if (p.div > 0.f) {
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) v[j][e] = v[j][e] / p.div;
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * (BN / 2) + j * 16 + g * 4;
        *reinterpret_cast<f32x4*>(p.C + (size_t)m * p.ldc + n) = v[j];
        if (p.C2) {
          f32x4 w2;
#pragma unroll
          for (int e = 0; e < 4; ++e) w2[e] = v[j][e] > 0.f ? v[j][e] : v[j][e] * p.c2_slope;
          *reinterpret_cast<f32x4*>(p.C2 + (size_t)m * p.ldc2 + n) = w2;
        }
      }
    }
  }
#endif
}

// ---- host side ---------------------------------------------------------------------------------
// Workspace, ticket counters and flags belong to the calling execution context (SkWorkspace, gemm.hpp).
// (XCD tile grouping: measured neutral on time and -6 % on L2 misses (profiles/r01_sk_sweep.txt), so off: Dispatch::sk_groups)
                              // tools: debug_force_tile(1, 8, g) switches it on
void conv_sk_set_groups(int on) { dispatch_edit([on](Dispatch& d) { d.sk_groups = on; }); }

bool conv_sk_eligible(const GemmArgs& a) {
  return a.same_rows && a.stride == 1 && a.chunk == 0 && !a.glu && !a.ln_g && a.Cin % SK_BK == 0 && (a.lda & 3) == 0 &&
         a.N % 64 == 0 && a.M > 0 && (a.ldc & 3) == 0 && (!a.R || (a.ldr & 3) == 0) && (!a.R2 || (a.ldr2 & 3) == 0) &&
         (!a.C2 || (a.ldc2 & 3) == 0) && ((size_t)(a.M + a.pad + 128) * a.lda + a.Cin) * 4 < 0x7ff00000ull &&
         (size_t)a.N * a.taps * a.Cin * 4 < 0x7ff00000ull && (a.in_act == ACT_NONE || (a.in_act == ACT_LRELU && a.in_slope > 0.f && a.in_slope < 1.f));
}

int conv_sk_error_count() { return sk_workspace_error_count(); }   // both stream-K generations share the workspaces

template <int BN, bool LRELU>
static int launch_sk(const GemmArgs& a, hipStream_t stream, int g_force) {
  constexpr size_t kLds = 2 * (size_t)(SK_BM + BN) * SK_BK * sizeof(float) + (2 * SK_BM + 4) * sizeof(int);
  SS_MAX_LDS_ONCE((&conv_sk_kernel<BN, LRELU>), kLds);
  SkWorkspace* st = nullptr;
  int rc = sk_workspace_acquire(stream, &st);
  if (rc != SS_OK) return rc;
  const int cus = 2 * st->cus > SK_MAXG ? SK_MAXG / 2 : st->cus;
  const long long nk = (long long)a.taps * (a.Cin / SK_BK);
  const long long U = (long long)cdiv(a.M, SK_BM) * (a.N / BN) * nk;
  // every workgroup gets >= 8 k-steps so that the fix-up (one 64-KB partial) stays a small fraction
  long long G = g_force > 0 ? g_force : 2LL * cus;
  if (G > U / 8) G = U / 8;
  if (G < 1) G = 1;
  if (G > SK_MAXG) G = SK_MAXG;
  if (G > 2LL * st->cus) G = 2LL * st->cus;               // the workspace holds 2 x 64 KB per CU
  const long long tiles = (long long)cdiv(a.M, SK_BM) * (a.N / BN);
  // data-parallel special case: when the tile count itself nearly fills the resident grid, one tile per
  // workgroup needs no fix-up at all (U/G = nk exactly)
  if (g_force <= 0 && tiles <= G && 4 * tiles >= 3 * G) G = tiles;
  const int NG = (disp().sk_groups && G >= 64 && tiles >= 64) ? 8 : 1;
  if (NG > 1) G -= G % NG;
  SkArgs q;
  q.ws = st->ws; q.sync = st->sync1;
  unsigned epoch = st->epoch1 + 1;
  if (epoch == 0) epoch = 1;                              // 0 is what a fresh flag holds
  q.epoch = epoch; q.G = (int)G; q.NG = NG;
  for (int x = 0; x < 8; ++x) q.base[x] = st->base1[x];
  ProfRec rec{}; bool prof = false;
  rc = prof_begin(a, stream, 15, rec, prof);
  if (rc != SS_OK) return rc;
  hipLaunchKernelGGL((conv_sk_kernel<BN, LRELU>), dim3((unsigned)G), dim3(256), kLds, stream, a, q);
  SS_LAUNCH_CHECK();
  // the launch is in the stream: only now do the tickets it will draw and its epoch become part of the context's state
  for (int x = 0; x < NG; ++x) st->base1[x] += (unsigned)(G / NG);
  st->epoch1 = epoch;
  return prof_end(stream, rec, prof);
}

int launch_conv_sk(const GemmArgs& a, hipStream_t stream, int g_force) {
  if (!conv_sk_eligible(a)) return SS_ERR_ARG;
  const bool lr = a.in_act == ACT_LRELU;
  if (a.N % 128 == 0) return lr ? launch_sk<128, true>(a, stream, g_force) : launch_sk<128, false>(a, stream, g_force);
  return lr ? launch_sk<64, true>(a, stream, g_force) : launch_sk<64, false>(a, stream, g_force);
}

}  // namespace ss
