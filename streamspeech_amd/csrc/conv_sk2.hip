// Persistent stream-K implicit-GEMM Conv1d, second generation (gfx950, exact-f32 MFMA): the N % 64 == 0 path of
// launch_conv_gemm -- HiFi-GAN resblock / up-sampling convs of a packed batch at C >= 64, conv_pre, the unit-decoder
// and encoder projections with enough k-steps (reference fairseq/models/text_to_speech/hifigan.py:52-172,
// SURVEY.md §8a rows a4, a12, a15).
//
// What changed against conv_sk.hip, and why (profiles/r01_sk_ablation.txt: of a 281-us launch the MFMA + ds_read stream
// alone took 211 us, LDS-DMA issue + barrier 38 us, fix-up + epilogue 32 us -- additive, because the two co-resident
// workgroups of a CU ran in phase; profiles/r02_sk2_ablation*.txt for this kernel):
//   * ONE workgroup per CU, 4 waves, one per SIMD, on a 256 x BN tile (BN = 128: wave tile 128 x 64 = 32 accumulator
//     tiles, 256 MFMAs per wave between barriers; BN = 64: wave tile 64 x 64, 128 MFMAs): 25 % fewer LDS-DMA pieces and
//     ds_reads per MFMA than 128 x 128 tiles, and nothing on a SIMD competes with its wave for issue slots.
//   * 3-stage LDS ring (3 x 48 | 40 KB): k-step c + 2 is staged while c is contracted, so the DMA has two full steps
//     (~16k cycles) to land; waits are counted (vmcnt(#pieces) leaves the newest step in flight across the barrier), the
//     barrier is a raw s_barrier (a __syncthreads would drain the DMA queue).
//   * the k-loop is software-pipelined inside the wave and its body is ONE basic block (no branches: the last steps of a
//     launch stage zero fills): the fragments of the second half-step are read under the first half's MFMAs, the first
//     fragments of the NEXT step under the second half's, and hipcc interleaves DMA pieces, ds_reads and MFMAs.
//     (Forcing "8 MFMAs, 1 ds_read, 1 piece" groups with sched_group_barrier or sched_barrier fences was measured
//     slower: the register allocator then slides the accumulators, ~100 v_accvgpr moves per step.)
//   * the stager runs across part boundaries: when a workgroup finishes a tile part, the next part's first two steps
//     are already in flight, the pipeline never drains.
//   * stream-K hand-off as in conv_sk.hip: ranges are walked last-tile-first, so the part a workgroup has to park (the
//     head k-steps of the last tile of its range) is done FIRST and the tile it has to finish comes LAST -- by then the
//     lower-numbered workgroups that share it have long parked theirs.  Logical workgroup ids come from an atomic
//     ticket, so a finisher only ever waits for workgroups that are already running (several stream-K launches on
//     different streams may share the chip).  Partials are summed in workgroup order: deterministic.
//   * epilogue: residual loads of 8 accumulator tiles in flight at a time, every option ONE wave-uniform branch around a
//     loop, SiLU / tanh out of line (a per-element switch with expf / tanhf inlined made a 115-KB kernel whose epilogue
//     thrashed the instruction cache and cost 25+ us per tile part).
#include "gemm.hpp"

#include <cstdlib>

namespace ss {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using s16x4 = __attribute__((ext_vector_type(4))) short;
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

constexpr int K2_BM = 256, K2_BK = 32;
[[maybe_unused]] constexpr int K2_SC1 = 16;                              // buffer cache policy bit: agent scope
constexpr int K2_MAXG = 512;
[[maybe_unused]] constexpr int K2_WORD0 = 16;                            // sync[K2_WORD0 + w] = flag of logical workgroup w
[[maybe_unused]] constexpr unsigned K2_SPIN_LIMIT = 1u << 22;
[[maybe_unused]] constexpr int K2_NUM_RECORDS = 0x7ffffff0;
[[maybe_unused]] constexpr unsigned K2_OOB = 0x80000000u;
constexpr int k2_stage_floats(int BN) { return (K2_BM + BN) * K2_BK; }
constexpr size_t k2_lds_bytes(int BN) { return 3 * (size_t)k2_stage_floats(BN) * sizeof(float) + (2 * K2_BM + 4) * sizeof(int); }

struct Sk2Args {
  float* ws;            // [G][256*BN] parked partial tiles, slot = logical workgroup id
  unsigned* sync;       // [0] ticket counter, [8] time-out counter, [K2_WORD0 + w] flag of logical workgroup w
  unsigned base;        // ticket value of logical workgroup 0 of this launch
  unsigned epoch;       // flag value meaning "the partial of this launch is in place"
  int G;                // workgroups (<= CUs: one per CU, all resident)
  unsigned long long* dbg;   // K2_TIMING builds only (tools/sk2_timing.py): per-workgroup phase cycle counts; nullptr otherwise
};

#define K2_WAIT_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
// Diagnostic hook points.  The product build compiles every one of them to nothing.  The timing-only ablation switches
// (results wrong by design) and the per-phase cycle stamps that produced profiles/r02_sk2_ablation*.txt and
// profiles/r02_sk2_timing*.txt live in tools/src/conv_sk2_diag.hpp, which is only included by the diagnostic library
// build (tools/sk2_bench.py / tools/sk2_timing.py: SS_EXTRA_FLAGS="-DK2_DIAGNOSTIC_BUILD -DK2_ABL=<mask>" or "-DK2_TIMING=1").
#ifdef K2_DIAGNOSTIC_BUILD
#include "../../tools/src/conv_sk2_diag.hpp"
#else
#define K2D_SKIP_DMA(in_loop) false
#define K2D_SKIP_A_PIECE(in_loop, j) false
#define K2D_SKIP_WAIT_BARRIER 0
#define K2D_SKIP_DS_READ(in_loop) false
#define K2D_SKIP_TAIL false
#define K2D_SKIP_MFMA false
#define K2D_SKIP_HANDOFF false
#define K2D_SKIP_STORES false
#define K2D_SKIP_R_LOADS false
#define K2D_ONE_A_PIECE 0
#define K2D_TIMING 0
#define K2D_TIMING_DECL
#define K2_STAMP(slot) do { } while (0)
#define K2D_COUNT_STEPS(n) do { } while (0)
#define K2D_TIMING_FLUSH do { } while (0)
#endif
#ifndef K2_RPREF
#define K2_RPREF 1      // 1: request the tile's residual operand during the part's last k-step (see the k-loop)
#endif

// SiLU / tanh epilogues are rare on this path (no vocoder conv uses them): out of line, see the header.
__device__ __attribute__((noinline)) float k2_act_slow(float v, int act) {
  if (act == ACT_SILU) return v / (1.0f + expf(-v));
  if (act == ACT_TANH) return tanhf(v);
  return v;
}

// Split-bf16 mode (X3, opt-in per vocoder, never the default): a fragment of 4 f32 values becomes {hi01, hi23, lo01, lo23}
// with hi = bf16(x) (round to nearest even) and lo = bf16(x - hi), i.e. x to 16 significant bits; the k-slice of 16 is
// then contracted by THREE v_mfma_f32_16x16x16_bf16 (W_hi.A_hi + W_hi.A_lo + W_lo.A_hi, f32 accumulate; 16 cycles each)
// instead of FOUR v_mfma_f32_16x16x4_f32 (32 cycles each).  The lane layout of the bf16 instruction (lane (r, g) holds
// k = 4g .. 4g+3 of row r) is exactly the layout of the f32x4 fragments, so nothing else in the kernel changes.
__device__ __forceinline__ f32x4 k2_split_bf16(f32x4 x) {
  const bf16x2_t h01 = __builtin_convertvector((f32x2){x[0], x[1]}, bf16x2_t);
  const bf16x2_t h23 = __builtin_convertvector((f32x2){x[2], x[3]}, bf16x2_t);
  const unsigned u01 = __builtin_bit_cast(unsigned, h01), u23 = __builtin_bit_cast(unsigned, h23);
  const bf16x2_t l01 = __builtin_convertvector((f32x2){x[0] - __uint_as_float(u01 << 16), x[1] - __uint_as_float(u01 & 0xffff0000u)}, bf16x2_t);
  const bf16x2_t l23 = __builtin_convertvector((f32x2){x[2] - __uint_as_float(u23 << 16), x[3] - __uint_as_float(u23 & 0xffff0000u)}, bf16x2_t);
  return f32x4{__uint_as_float(u01), __uint_as_float(u23), __uint_as_float(__builtin_bit_cast(unsigned, l01)),
               __uint_as_float(__builtin_bit_cast(unsigned, l23))};
}
__device__ __forceinline__ s16x4 k2_half(f32x4 v, int lo) {      // the hi (lo = 0) or lo (lo = 1) four bf16 of a split fragment
  const unsigned long long w = (unsigned long long)__float_as_uint(v[2 * lo]) | ((unsigned long long)__float_as_uint(v[2 * lo + 1]) << 32);
  return __builtin_bit_cast(s16x4, w);
}

template <int BN, bool LRELU, bool X3>
__global__ __launch_bounds__(256, 1) void conv_sk2_kernel(const GemmArgs p, const Sk2Args q) {
#if __HIP_DEVICE_COMPILE__
  constexpr int BM = K2_BM, BK = K2_BK, STAGE = k2_stage_floats(BN);
  constexpr int TN = 4, TM = BN == 128 ? 8 : 4;  // 16x16 MFMA tiles per wave: wave tile 128 x 64 (BN = 128) | 64 x 64 (BN = 64)
  constexpr int NWP = BN / 32;                   // W pieces per wave per step (a piece = 8 rows x 128 B = one LDS-DMA instruction)
  constexpr int PIECES = 8 + NWP;                // DMA pieces per wave per step: 64 A rows + BN / 4 W rows
  constexpr int NFR = TM + TN;                   // b128 fragments per half-step
  constexpr int HG = TM / 2;                     // groups of 8 MFMAs per k4 slice
  static_assert(BN == 128 || BN == 64, "tile widths");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  int* s_lo = reinterpret_cast<int*>(smem + 3 * STAGE);
  int* s_hi = s_lo + BM;
  int* s_misc = s_hi + BM;

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wrow = BN == 128 ? (wave >> 1) * 128 : wave * 64;   // first tile row / column of this wave
  const int wcol = BN == 128 ? (wave & 1) * 64 : 0;
  const int r = lane & 15, g = lane >> 4;
  if (t == 0) s_misc[0] = (int)(atomicAdd(q.sync, 1u) - q.base);
  __syncthreads();
  const int w = __builtin_amdgcn_readfirstlane(s_misc[0]);   // logical id in order of arrival: range, workspace slot, flag
  // a ticket outside [0, G) means the host's ticket base and the device counter disagree (two host threads driving one
  // context): count it like a bounded-wait time-out and leave, instead of indexing tiles / workspace slots with it
  if ((unsigned)w >= (unsigned)q.G) { if (t == 0) atomicAdd(q.sync + 8, 1u); return; }
  K2D_TIMING_DECL

  const int kpt = p.Cin / BK;
  const int nk = p.taps * kpt;
  const int Ktot = p.taps * p.Cin;
  const int tiles_n = p.N / BN;
  const int tiles_m = (p.M + BM - 1) / BM;
  const long long U = (long long)tiles_m * tiles_n * nk;
  const long long u0 = (long long)w * U / q.G, u1 = (long long)(w + 1) * U / q.G;
  if (u1 <= u0) return;
  const int t_first = (int)(u0 / nk), t_last = (int)((u1 - 1) / nk);

  const float slope = p.in_slope;
  // Buffer resources of the LDS-DMA loads: per-lane byte offsets stay constant over a tile part, the k-step
  // (tap shift, channel block) goes into the wave-uniform soffset; rows outside their utterance get an offset
  // beyond num_records and the range check returns zeros (the conv's zero padding).  The A base is moved back by
  // `pad` rows so that every valid offset is non-negative.
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(reinterpret_cast<uintptr_t>(p.A) - (uintptr_t)p.pad * p.lda * sizeof(float)), 0, K2_NUM_RECORDS, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, K2_NUM_RECORDS, 0x00020000);

  // per-lane LDS read offsets (floats) inside a stage: 16-B chunk c of row rr sits at chunk position c ^ ((rr >> 1) & 7)
  // (the swizzle is applied to the DMA's SOURCE address; rows of a 16-row MFMA tile differ only in r)
  const int swz = (r >> 1) & 7;
  const int rdA0 = (wrow + r) * BK + ((g ^ swz) << 2);
  const int rdA1 = (wrow + r) * BK + (((4 + g) ^ swz) << 2);
  const int rdW0 = BM * BK + (wcol + r) * BK + ((g ^ swz) << 2);
  const int rdW1 = BM * BK + (wcol + r) * BK + (((4 + g) ^ swz) << 2);
  const int st_row = lane >> 3, st_pos = lane & 7;       // DMA piece = 8 rows x 8 chunks

  // ---- state of the part whose k-steps are being staged (registers; the k-step scalars are wave-uniform) ----
  struct Part { int ka, kb, n, m0, n0, tm; };
  const int nparts = t_last - t_first + 1;
  auto part_of = [&](int ip) {
    const int tile = t_last - ip;              // last tile first: see the header
    const long long ut0 = (long long)tile * nk;
    Part P;
    P.ka = (int)(max(u0, ut0) - ut0);
    P.kb = (int)(min(u1, ut0 + nk) - ut0);
    P.n = P.kb - P.ka;
    P.tm = tile / tiles_n;
    P.m0 = P.tm * BM;
    P.n0 = (tile - P.tm * tiles_n) * BN;
    return P;
  };
  int a_base[8], a_len[8];              // row of piece j relative to its utterance (at tap shift 0), rows in that utterance
  unsigned a_off[8], w_off[NWP];
  int nci = 0, ntap = 0;                 // of the next step to stage
  int soffA = 0, soffW = 0, shift = 0;
  int cur_tm = -1;
  // Row bounds + per-lane offsets of a part.  Called when every wave is past the previous part's reads of them (two
  // barriers inside when the row tile changes).
  // Barriers here are raw s_barriers behind an LDS-only wait: a __syncthreads() would also wait for vmcnt(0), i.e. drain
  // the two k-steps of LDS-DMA that are in flight while the next part is being set up (1-2 us per tile part).
  auto lds_barrier = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  };
  auto stage_setup = [&](const Part& P) {
    lds_barrier();
    if (P.tm != cur_tm) {
      cur_tm = P.tm;
      const int m = P.m0 + t;            // 256 threads, 256 rows
      int lo = 0, hi = 0;
      if (m < p.M) {
        if (p.nseg > 0) {
          // segments are contiguous and ascending (same_rows): binary search for the one that holds row m
          // (a linear scan is nseg dependent scalar loads -- ~3 us per row tile at 32 utterances)
          int a = 0, bsz = p.nseg;
          while (bsz > 1) {
            const int half = bsz >> 1;
            if (p.segs[4 * (a + half)] <= m) a += half;
            bsz -= half;
          }
          const int sst = p.segs[4 * a], ln = p.segs[4 * a + 1];
          if (m >= sst && m < sst + ln) { lo = sst; hi = sst + ln; }
        } else {
          hi = p.in_len;
        }
      }
      s_lo[t] = lo; s_hi[t] = hi;
      lds_barrier();
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int row = wave * 64 + j * 8 + st_row;
      a_base[j] = P.m0 + row - p.pad - s_lo[row];
      a_len[j] = s_hi[row] - s_lo[row];             // 0 for rows outside every utterance: the range test below always fails
      a_off[j] = (unsigned)(((P.m0 + row) * p.lda + ((st_pos ^ ((row >> 1) & 7)) << 2)) * 4);
    }
#pragma unroll
    for (int j = 0; j < NWP; ++j) {
      const int nrow = wave * (BN / 4) + j * 8 + st_row;
      w_off[j] = (unsigned)(((P.n0 + nrow) * Ktot + ((st_pos ^ ((nrow >> 1) & 7)) << 2)) * 4);
    }
    // k-step s = (channel block s / taps, tap s % taps): the taps of one 32-channel block run back to back, so the
    // (256 + halo) x 128-B slab of A rows they share is re-read from L1/L2 while it is still there
    nci = (P.ka / p.taps) * BK;
    ntap = P.ka - (P.ka / p.taps) * p.taps;
  };
  int shift_l = 0;                       // tap shift for the row-range test; beyond every utterance when nothing is staged
  auto step_begin = [&](bool live) {     // scalars of the step about to be staged.  readfirstlane: they MUST be SGPRs --
    const int tapv = __builtin_amdgcn_readfirstlane(ntap);   // a soffset the compiler keeps in a VGPR turns every DMA piece
    const int nciv = __builtin_amdgcn_readfirstlane(nci);    // into a waterfall loop (cdna_hip_programming.md T20)
    shift = tapv * p.dil;
    shift_l = live ? shift : 0x40000000;
    soffA = (shift * p.lda + nciv) * 4;
    soffW = (tapv * p.Cin + nciv) * 4;
  };
  auto step_advance = [&](bool adv = true) {   // branch-free: the k-loop body must stay ONE basic block
    const int nt = ntap + (adv ? 1 : 0);
    const bool wrap = nt >= p.taps;
    ntap = wrap ? 0 : nt;
    nci += wrap ? BK : 0;
  };
  // `live` = false turns a piece into a zero fill (offset beyond the buffer range: no memory traffic): when nothing is
  // left to stage the k-loop keeps issuing into ring slots nobody reads any more, so its body has no branches and the
  // wait count is the same every iteration.
  [[maybe_unused]] bool in_loop = false;          // read by the diagnostic hooks only
  auto issueA = [&](int stage, int j, bool live) {
    if (K2D_SKIP_DMA(in_loop) || K2D_SKIP_A_PIECE(in_loop, j)) return;
    float* sA = smem + stage * STAGE + (wave * 64) * BK;
    const bool ok = (unsigned)(a_base[j] + shift_l) < (unsigned)a_len[j];   // one unsigned compare: row inside its utterance (and live)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr_t)(sA + j * 8 * BK), 16, ok ? a_off[j] : K2_OOB,
                                             __builtin_amdgcn_readfirstlane(soffA), 0, 0);
  };
  auto issueW = [&](int stage, int j, bool live) {
    if (K2D_SKIP_DMA(in_loop)) return;
    float* sW = smem + stage * STAGE + BM * BK + (wave * (BN / 4)) * BK;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lds_ptr_t)(sW + j * 8 * BK), 16, live ? w_off[j] : K2_OOB,
                                             __builtin_amdgcn_readfirstlane(soffW), 0, 0);
  };
  auto issue_step = [&](int stage, bool live) {
    step_begin(live);
#pragma unroll
    for (int j = 0; j < 8; ++j) issueA(stage, j, live);
#pragma unroll
    for (int j = 0; j < NWP; ++j) issueW(stage, j, live);
    step_advance(live);
  };
  auto wait_newest_step_only = [&]() {   // everything but the PIECES newest VMEM operations of this wave has completed
    if constexpr (K2D_ONE_A_PIECE != 0) { if constexpr (BN == 128) K2_WAIT_VMCNT(5); else K2_WAIT_VMCNT(3); }
    else if constexpr (BN == 128) K2_WAIT_VMCNT(12); else K2_WAIT_VMCNT(10);
  };
  // ---- the stager: runs two k-steps ahead of the contraction, straight across part boundaries ----
  // All k-steps of all parts of this workgroup form one sequence; step c is contracted out of ring slot c % 3 while
  // step c + 2 is staged.  When the stager finishes a part it moves on to the next one (row bounds + offsets), so at a
  // part boundary the next part's first two steps are already in flight / landed.
  int stage_ip = 0, stage_left = 0;      // part being staged, steps of it still to stage
  auto stager_next_part = [&]() {
    const Part P = part_of(stage_ip);
    stage_setup(P);
    stage_left = P.n;
  };
  auto stager_advance = [&]() {          // wave-uniform, taken once per part
    if (stage_left == 0 && stage_ip + 1 < nparts) { ++stage_ip; stager_next_part(); }
  };

  int st = 0;                            // ring slot of the step being contracted (runs on across parts)
  stager_next_part();
  issue_step(0, true);
  --stage_left;
  stager_advance();
  issue_step(1, stage_left > 0);
  stage_left -= stage_left > 0 ? 1 : 0;
  wait_newest_step_only();
  __builtin_amdgcn_s_barrier();
  K2_STAMP(0);
  for (int ip = 0; ip < nparts; ++ip) {
    const Part cur = part_of(ip);
    const int n = cur.n, m0 = cur.m0, n0 = cur.n0;
    K2D_COUNT_STEPS(n);
    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto load_frag = [&](const float* S, int half, int u, f32x4 (&af)[TM], f32x4 (&bf)[TN]) {   // fragment u of a half-step: B 0..3, A 4..
      if (K2D_SKIP_DS_READ(in_loop)) return;
      if (u < TN) {
        bf[u] = *reinterpret_cast<const f32x4*>(S + (half ? rdW1 : rdW0) + u * 16 * BK);
        if (X3) bf[u] = k2_split_bf16(bf[u]);
      } else {
        const int i = u - TN;
        af[i] = *reinterpret_cast<const f32x4*>(S + (half ? rdA1 : rdA0) + i * 16 * BK);
        if (LRELU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) af[i][e] = fmaxf(af[i][e], af[i][e] * slope);
        }
        if (X3) af[i] = k2_split_bf16(af[i]);
      }
    };
    // 8 MFMAs: k4 slice e of row tiles 2q, 2q+1 against all 4 column tiles (8 independent accumulators; the same
    // accumulator comes back 4 * TM MFMAs later)
    auto mma8 = [&](const f32x4 (&af)[TM], const f32x4 (&bf)[TN], int e, int qd) {
      if (K2D_SKIP_MFMA) return;
      if (X3 && e == 3) return;              // split-bf16: three terms per k-slice of 16 (e = 0: hi.hi, 1: hi.lo, 2: lo.hi)
#pragma unroll
      for (int i = 2 * qd; i < 2 * qd + 2; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          if constexpr (X3)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(k2_half(bf[j], e == 2), k2_half(af[i], e == 1), acc[i][j], 0, 0, 0);
          else
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[j][e], af[i][e], acc[i][j], 0, 0, 0);   // D = W.A^T: see epilogue
        }
    };

    // The part's first step landed and was made visible by the previous part's last barrier (the initial one for part 0);
    // its first fragments are (re)read here rather than kept in registers across the epilogue (VGPRs the epilogue needs).
    f32x4 ax[TM], bx[TN], ay[TM], by[TN];
#pragma unroll
    for (int u = 0; u < NFR; ++u) load_frag(smem + st * STAGE, 0, u, ax, bx);
#if K2_RPREF
    // The residual operand R of the WHOLE tile (TM x TN float4 per lane) is requested at the top of the part's last
    // k-step, so that it has landed when the epilogue starts: with one workgroup per CU nothing else hides that latency.
    // (The loads sit in the VMEM queue between two steps' DMA pieces; the counted wait then simply covers them too.)
    constexpr int PT = BN == 128 ? 2 : TM;                     // row tiles requested ahead (the 128-wide variant has no registers for more)
    f32x4 rall[PT][TN];
    const bool want_r = p.R != nullptr && cur.kb == nk;        // only the finisher of a tile needs R
#endif
    in_loop = true;
    K2_STAMP(1);
    for (int i = 0; i < n; ++i) {
#if K2_RPREF
      if (i == n - 1 && want_r) {
        int lp = lane;
        asm volatile("" : "+v"(lp));
#pragma unroll
        for (int ii = 0; ii < PT; ++ii) {
          const int m = min(m0 + wrow + ii * 16 + (lp & 15), p.M - 1);
#pragma unroll
          for (int j = 0; j < TN; ++j)
            rall[ii][j] = *reinterpret_cast<const f32x4*>(p.R + (size_t)m * p.ldr + n0 + wcol + j * 16 + (lp >> 4) * 4);
        }
      }
#endif
      stager_advance();
      const float* S = smem + st * STAGE;
      const int st1 = st == 2 ? 0 : st + 1, st2 = st1 == 2 ? 0 : st1 + 1;
      const bool live = stage_left > 0;          // nothing left to stage: zero fills (the count stays PIECES per step)
      stage_left -= live ? 1 : 0;
      const float* S1 = smem + st1 * STAGE;      // (after the very last step: a slot nobody reads from again)
      // ---- contract X (first half-step) and k4 slices 0-1 of Y; read Y; stage step c + 2 ----
      // (source order: a ds_read may not be moved across an LDS-DMA piece -- both touch LDS -- so they alternate)
      step_begin(live);
#pragma unroll
      for (int u = 0; u < (NFR > PIECES ? NFR : PIECES); ++u) {
        if (u < NFR) load_frag(S, 1, u, ay, by);
        if (u < 8) issueA(st2, u, live); else if (u < PIECES) issueW(st2, u - 8, live);
      }
      step_advance(live);
#pragma unroll
      for (int sl = 0; sl < 4 * HG; ++sl) mma8(ax, bx, sl / HG, sl % HG);
#pragma unroll
      for (int sl = 0; sl < 2 * HG; ++sl) mma8(ay, by, sl / HG, sl % HG);
      // ---- step c + 1 landed for every wave, every wave is done with slot st - 1 and with X ----
#if !K2D_SKIP_WAIT_BARRIER
      wait_newest_step_only();
      __builtin_amdgcn_s_barrier();
#endif
      // ---- k4 slices 2-3 of Y; the first-half fragments of step c + 1 are read under them ----
#pragma unroll
      for (int u = 0; u < NFR; ++u) load_frag(S1, 0, u, ax, bx);
#pragma unroll
      for (int sl = 0; sl < 2 * HG; ++sl) mma8(ay, by, 2 + sl / HG, sl % HG);
      st = st1;
    }
    in_loop = false;
    K2_STAMP(2);
    const bool has_begin = cur.ka == 0, has_end = cur.kb == nk;
    if (K2D_SKIP_TAIL && acc[0][0][0] != 12345.f) continue;

    // Everything below derives its per-lane addresses from `le`, which the compiler cannot see through: otherwise it
    // computes the hand-off / epilogue addresses BEFORE the k-loop and carries ~100 VGPRs through it.
    int le = lane;
    asm volatile("" : "+v"(le));
    const int r_e = le & 15, g_e = le >> 4;
    if (!K2D_SKIP_HANDOFF && !has_end) {
      // ---- contributor: park the partial tile, raise the flag ----
      // Partials move as sc1 (agent-scope) b128 buffer stores / loads and the flags as sc1 relaxed atomics: they write
      // through / read past the per-XCD L2, so no L2 write-back or invalidate (which would evict the weights every
      // other workgroup of the XCD is streaming) is needed.  Order: stores complete (vmcnt 0) -> workgroup barrier -> flag.
      const __amdgpu_buffer_rsrc_t rsP = __builtin_amdgcn_make_buffer_rsrc((void*)(q.ws + (size_t)w * (BM * BN)), 0, BM * BN * 4, 0x00020000);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          u32x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = __float_as_uint(acc[i][j][e]);
          __builtin_amdgcn_raw_buffer_store_b128(v, rsP, (((wave * TM + i) * TN + j) * 64 + le) * 16, 0, K2_SC1);
        }
      K2_WAIT_VMCNT(0);
      __syncthreads();
      if (t == 0) __hip_atomic_store(q.sync + K2_WORD0 + w, q.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      K2_STAMP(3);
      continue;
    }
    if (!K2D_SKIP_HANDOFF && !has_begin) {
      // ---- finisher: add the partials of the workgroups that own k-steps [0, ka) of this tile, in workgroup order ----
      const long long ut0 = (long long)(t_last - ip) * nk;
      const int wf = (int)(((ut0 + 1) * q.G - 1) / U);       // workgroup that owns the tile's first unit
      if (t == 0) {
        for (int ww = wf; ww < w; ++ww) {
          unsigned spins = 0;
          while (__hip_atomic_load(q.sync + K2_WORD0 + ww, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != q.epoch) {
            __builtin_amdgcn_s_sleep(8);
            if (++spins > K2_SPIN_LIMIT) { atomicAdd(q.sync + 8, 1u); break; }
          }
        }
      }
      __syncthreads();
      for (int ww = wf; ww < w; ++ww) {        // fixed order: ((mine + P[wf]) + P[wf+1]) + ...
        const __amdgpu_buffer_rsrc_t rsP = __builtin_amdgcn_make_buffer_rsrc((void*)(q.ws + (size_t)ww * (BM * BN)), 0, BM * BN * 4, 0x00020000);
        // a parked part comes in groups of 8 tiles, all 8 loads of a group in flight before the first add
#pragma unroll
        for (int qq = 0; qq < HG; ++qq) {
          __builtin_amdgcn_sched_barrier(0);
          u32x4 o[2][TN];
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
              o[i][j] = __builtin_amdgcn_raw_buffer_load_b128(rsP, (((wave * TM + qq * 2 + i) * TN + j) * 64 + le) * 16, 0, K2_SC1);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
              for (int e = 0; e < 4; ++e) acc[qq * 2 + i][j][e] += __uint_as_float(o[i][j][e]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }

    K2_STAMP(3);
    // ---- epilogue (same operation order as conv_gemm_kernel / conv_sk_kernel) ----
    // The MFMAs were issued with the operands swapped (D = W_tile . A_tile^T), so in the C/D layout
    // (col = lane & 15, row = 4 * (lane >> 4) + reg) a lane holds 4 CONSECUTIVE output channels of ONE row:
    // bias / residual loads and the stores are float4.  Needs ldc/ldr/ldr2/ldc2 % 4 == 0 (checked on the host).
    // In groups of 8 accumulator tiles: ALL residual loads of a group (R and R2: up to 16 KB per wave) are issued
    // before its first use -- with one workgroup per CU nothing else hides the memory latency.  sched_barrier(0) on
    // both sides keeps the compiler from merging groups (which spills).
    if (m0 + wrow >= p.M) continue;
    f32x4 bb[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      bb[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (p.bias) bb[j] = *reinterpret_cast<const f32x4*>(p.bias + n0 + wcol + j * 16 + g_e * 4);
    }
#pragma unroll
    for (int qq = 0; qq < HG; ++qq) {
      __builtin_amdgcn_sched_barrier(0);
      f32x4 rr[2][TN], rr2[2][TN];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int m = min(m0 + wrow + (qq * 2 + i) * 16 + r_e, p.M - 1);    // clamped: rows >= M are never stored
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int nn = n0 + wcol + j * 16 + g_e * 4;
#if K2_RPREF
          if (p.R && !K2D_SKIP_R_LOADS) {
            if (qq * 2 + i < PT) rr[i][j] = rall[qq * 2 + i < PT ? qq * 2 + i : 0][j];
            else rr[i][j] = *reinterpret_cast<const f32x4*>(p.R + (size_t)m * p.ldr + nn);
          }
#else
          if (p.R && !K2D_SKIP_R_LOADS) rr[i][j] = *reinterpret_cast<const f32x4*>(p.R + (size_t)m * p.ldr + nn);
#endif
          if (p.R2 && !K2D_SKIP_R_LOADS) rr2[i][j] = *reinterpret_cast<const f32x4*>(p.R2 + (size_t)m * p.ldr2 + nn);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      // every option is ONE wave-uniform branch around a loop over the group's 32 elements (not a branch per element)
      f32x4 v[2][TN];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          v[i][j] = acc[qq * 2 + i][j];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[i][j][e] += bb[j][e];
        }
      if (p.act == ACT_LRELU) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) v[i][j][e] = v[i][j][e] > 0.f ? v[i][j][e] : v[i][j][e] * p.act_slope;
      } else if (p.act == ACT_RELU) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) v[i][j][e] = fmaxf(v[i][j][e], 0.f);
      } else if (p.act != ACT_NONE) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) v[i][j][e] = k2_act_slow(v[i][j][e], p.act);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) v[i][j][e] *= p.alpha;
      if (p.R && !K2D_SKIP_R_LOADS) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) v[i][j][e] += rr[i][j][e];
      }
      if (p.R2 && !K2D_SKIP_R_LOADS) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) v[i][j][e] = rr2[i][j][e] + v[i][j][e];
      }
      if (p.div > 0.f) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) v[i][j][e] = v[i][j][e] / p.div;
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int m = m0 + wrow + (qq * 2 + i) * 16 + r_e;
        if (m >= p.M) continue;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int nn = n0 + wcol + j * 16 + g_e * 4;
          if (!K2D_SKIP_STORES || v[i][j][0] == 12345.f) *reinterpret_cast<f32x4*>(p.C + (size_t)m * p.ldc + nn) = v[i][j];
          if (p.C2 && !K2D_SKIP_STORES) {
            f32x4 w2;
#pragma unroll
            for (int e = 0; e < 4; ++e) w2[e] = v[i][j][e] > 0.f ? v[i][j][e] : v[i][j][e] * p.c2_slope;
            *reinterpret_cast<f32x4*>(p.C2 + (size_t)m * p.ldc2 + nn) = w2;
          }
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    K2_STAMP(4);
  }
  K2D_TIMING_FLUSH;
#endif
}

// ---- host side ---------------------------------------------------------------------------------
// Workspace, ticket counter and flags belong to the calling execution context (SkWorkspace, gemm.hpp).
#if K2D_TIMING
static unsigned long long* g_k2_last_dbg = nullptr;
static int g_k2_last_G = 0;
// diagnostic build only: phase cycle counts of the LAST conv_sk2 launch, [G][8] = prologue, part set-up, k-loop, hand-off,
// epilogue, k-steps, t_begin, t_end (s_memtime cycles; slots are per workgroup in ticket order)
extern "C" int ss_debug_sk2_timing(unsigned long long* h_out, int cap_wgs) {
  if (!g_k2_last_dbg || cap_wgs < g_k2_last_G) return -1;
  if (hipMemcpy(h_out, g_k2_last_dbg, (size_t)g_k2_last_G * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) return -1;
  return g_k2_last_G;
}
#endif
static const int g_k2_spare_cus = getenv("SS_SK2_SPARE_CUS") ? atoi(getenv("SS_SK2_SPARE_CUS")) : 0;   // tuning knob (measured flat)

// (Rounds 4-5 had a pack-invariant mode here -- ranges cut on whole tiles, one accumulator chain over all of K.  Round 6's blocked
//  chain needs a second accumulator set, which 128 accumulator registers per wave leave no room for: CANON_SEQ launches no longer
//  come here -- launch_canon in gemm.hip -- and a.canon is refused below.)
bool conv_sk2_eligible(const GemmArgs& a) {
  return a.canon != CANON_SEQ && a.same_rows && a.stride == 1 && a.chunk == 0 && !a.glu && !a.ln_g && a.Cin % K2_BK == 0 && (a.lda & 3) == 0 &&
         a.N % 64 == 0 && a.M > 0 && (a.ldc & 3) == 0 && (!a.R || (a.ldr & 3) == 0) && (!a.R2 || (a.ldr2 & 3) == 0) &&
         (!a.C2 || (a.ldc2 & 3) == 0) && ((size_t)(a.M + a.pad + 256) * a.lda + a.Cin) * 4 < 0x7ff00000ull &&
         (size_t)a.N * a.taps * a.Cin * 4 < 0x7ff00000ull &&
         (a.in_act == ACT_NONE || (a.in_act == ACT_LRELU && a.in_slope > 0.f && a.in_slope < 1.f));
}

int conv_sk2_error_count() { return 0; }   // counted with the shared workspaces: conv_sk_error_count()

template <int BN, bool LRELU, bool X3>
static int launch_sk2(const GemmArgs& a, hipStream_t stream, int g_force) {
  constexpr size_t kLds = k2_lds_bytes(BN);
  SS_MAX_LDS_ONCE((&conv_sk2_kernel<BN, LRELU, X3>), kLds);
  SkWorkspace* st = nullptr;
  int rc = sk_workspace_acquire(stream, &st);
  if (rc != SS_OK) return rc;
  int cus = st->cus > K2_MAXG ? K2_MAXG : st->cus;
  // SS_SK2_SPARE_CUS = n: leave n CUs to the other streams' small kernels (a stream-K workgroup owns its CU: 480 of
  // the 512 registers per SIMD, 147 KB of LDS -- nothing can be co-resident with it)
  if (g_k2_spare_cus > 0 && g_k2_spare_cus < cus) cus -= g_k2_spare_cus;
  const long long nk = (long long)a.taps * (a.Cin / K2_BK);
  const long long U = (long long)cdiv(a.M, K2_BM) * (a.N / BN) * nk;
  long long G = g_force > 0 ? g_force : cus;        // one workgroup per CU (147 | 123 KB of LDS each), all resident
  if (G > cus) G = cus;
  if (G > U / 4) G = U / 4;                         // at least 4 k-steps per workgroup
  if (G < 1) G = 1;
  Sk2Args q;
  q.ws = st->ws; q.sync = st->sync2; q.G = (int)G; q.dbg = nullptr;
#if K2D_TIMING
  if (!st->dbg) SS_HIP_CHECK(hipMalloc(&st->dbg, (size_t)K2_MAXG * 8 * sizeof(unsigned long long)));
  SS_HIP_CHECK(hipMemsetAsync(st->dbg, 0, (size_t)K2_MAXG * 8 * sizeof(unsigned long long), stream));
  q.dbg = st->dbg;
  g_k2_last_dbg = st->dbg; g_k2_last_G = (int)G;
#endif
  unsigned epoch = st->epoch2 + 1;
  if (epoch == 0) epoch = 1;                        // 0 is what a fresh flag holds
  q.base = st->base2;
  q.epoch = epoch;
  ProfRec rec{}; bool prof = false;
  rc = prof_begin(a, stream, X3 ? 19 : 18, rec, prof);
  if (rc != SS_OK) return rc;
  hipLaunchKernelGGL((conv_sk2_kernel<BN, LRELU, X3>), dim3((unsigned)G), dim3(256), kLds, stream, a, q);
  SS_LAUNCH_CHECK();
  // the launch is in the stream: only now do the tickets it will draw and its epoch become part of the context's state
  st->base2 += (unsigned)G;
  st->epoch2 = epoch;
  return prof_end(stream, rec, prof);
}

int launch_conv_sk2(const GemmArgs& a, hipStream_t stream, int g_force) {
  if (!conv_sk2_eligible(a)) return SS_ERR_ARG;
  const bool lr = a.in_act == ACT_LRELU;
  if (a.x3) {     // opt-in split-bf16 contraction (ss_vocoder_set_bf16x3): never taken by the default f32 path
    if (a.N % 128 == 0) return lr ? launch_sk2<128, true, true>(a, stream, g_force) : launch_sk2<128, false, true>(a, stream, g_force);
    return lr ? launch_sk2<64, true, true>(a, stream, g_force) : launch_sk2<64, false, true>(a, stream, g_force);
  }
  if (a.N % 128 == 0) return lr ? launch_sk2<128, true, false>(a, stream, g_force) : launch_sk2<128, false, false>(a, stream, g_force);
  return lr ? launch_sk2<64, true, false>(a, stream, g_force) : launch_sk2<64, false, false>(a, stream, g_force);
}

}  // namespace ss
