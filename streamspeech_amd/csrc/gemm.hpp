// Implicit-GEMM Conv1d / Linear on the f32 matrix cores (v_mfma_f32_16x16x4_f32).
//
// One kernel family serves every dense contraction on the S2ST path (SURVEY.md §8a rows
// a2,a4,a5,a6,a8,a9,a11,a12,a14,a15): activations are time-major channels-last [rows, C];
//   out[m, n] = epi( sum_{j<taps} sum_{c<Cin} act_in(A[m*stride + j*dil - pad, c]) * W[n, j*Cin + c] )
// Linear layers are taps=1.  Conv1d weights are re-laid out tap-major [Cout][k][Cin] at pack
// time; ConvTranspose1d is packed as a 3-tap polyphase conv with N = stride*Cout
// (streamspeech_amd/weights.py).  Rows outside [0, in_len) read as zero ("same" padding); with
// chunk > 0 rows at or beyond ((m*stride)/chunk + 1)*chunk also read as zero, which is the
// closed form of the reference ChunkCausalConv1d (chunk_unity/modules/chunk_causal_conv1d.py:39-68).
#pragma once
#include "common.hpp"
#include "dispatch.hpp"

namespace ss {

struct GemmArgs {
  const float* A = nullptr;   // input rows, row stride lda
  const float* W = nullptr;   // [N][taps*Cin]
  const float* Wwino = nullptr;  // optional: the same weights in Winograd F(2,3) form [N][ceil(taps/3)*4*Cin] (conv_c64w.hip; made by launch_wino_pack)
  const float* bias = nullptr;  // [N] or null
  const float* R = nullptr;   // residual [M][ldr] or null
  const float* R2 = nullptr;  // second residual (MRF accumulate) or null
  float* C = nullptr;         // [M][ldc]
  int lda = 0, ldc = 0, ldr = 0, ldr2 = 0;
  int M = 0, N = 0, Cin = 0;
  int taps = 1, dil = 1, stride = 1, pad = 0;
  int in_len = 0;             // valid input rows
  int chunk = 0;              // chunk-causal visibility (0 = off)
  int m_begin = 0;            // conv_gemm tile kernel only (single utterance, no LayerNorm fusion): compute output rows [m_begin, M) -- pointers,
                              // in_len and the chunk rule stay those of row 0, so a row's bits do not depend on where the launch starts
                              // (the incremental streaming encoder re-runs the subsampler for its non-final rows only)
  int in_act = ACT_NONE;      // ACT_NONE or ACT_LRELU applied to A while staging
  float in_slope = 0.1f;
  int act = ACT_NONE;         // epilogue activation on (acc + bias)
  float alpha = 1.0f;         // v *= alpha (after activation)
  float div = 0.0f;           // if > 0: v /= div at the very end (MRF mean)
  float act_slope = 0.1f;     // slope of an ACT_LRELU epilogue activation
  // Optional second output C2 = leaky_relu(C, c2_slope): lets the *consumer* conv read a
  // pre-activated input (in_act = ACT_NONE) instead of applying the activation to every A fragment
  // inside its MFMA loop (VALU issue slots there cost matrix-core time one for one).
  float* C2 = nullptr;
  int ldc2 = 0;
  float c2_slope = 0.1f;
  int glu = 0;                // 1: W rows are interleaved [16 value | 16 gate] blocks, out has N/2 cols
  // ragged batch: nseg > 0 -> segs[4*s] = {out_start, out_len, in_start, in_len}; M/in_len ignored
  const int* segs = nullptr;
  int nseg = 0;
  int max_seg_out = 0;        // max out_len over segments (grid sizing)
  double algo_flops = 0.0;    // algorithmic FLOPs of this launch for the profiler (0 -> 2*M*N*taps*Cin)
  double algo_bytes = 0.0;    // algorithmic bytes of this launch for the profiler (0 -> weights + in + out + residuals once)
  // fused LayerNorm prologue on the A rows (linear layers only: taps == 1, normalised over Cin):
  // A' = (A - mean) * rstd * ln_g + ln_b, eps 1e-5 -- saves the separate LayerNorm launch.
  const float* ln_g = nullptr;
  const float* ln_b = nullptr;
  float* ln_out = nullptr;    // gemv path only (M <= 4): the normalised rows are also written here (row stride Cin) -- the MT
                              // decode step's final LayerNorm feeds both the vocabulary projection and the returned features
  // 1: the caller guarantees a "same" row mapping -- stride 1, output row m reads input rows
  // m - pad + j*dil of the same packed buffer, segments (if any) are contiguous with
  // out_start == in_start and out_len == in_len.  Makes the launch eligible for the persistent
  // stream-K kernel (conv_sk.hip).
  int same_rows = 0;
  int x3 = 0;                 // 1: split-bf16 (3 x bf16 MFMA) contraction where the kernel has one (conv_sk2 only; opt-in, see ss_vocoder_set_bf16x3)
  // Pack-invariant arithmetic: the bits of output row m must be a function of row m's operands alone -- not of
  // the row count M, the grid, or where stream-K cut the k-range.  CANON_SEQ: the k walk (16-wide slabs ascending, MFMA e of a slab
  // contracts k = {e, 4+e, 8+e, 12+e}) is cut into blocks of CANON_KBLOCK = 64; inside a block ONE accumulator chain, the block sums
  // are added in ascending order into a running total (first block: the total IS the block sum).  conv_gemm_kernel<.., BLK = true>
  // (no split-K), rt_linear (K = 256) and rt_linear_kb (K = 512 ...) implement exactly this -- the same bits
  // (tests/test_pack_invariance_gpu.py) -- so the choice among them may depend on M.  Rounds 4-5 ran ONE chain over all of K: at
  // K = 256 that is 1.35-1.5x, at K = 2048 2.8x farther from float64 than torch's CPU sgemm on the GPU box's host (VERDICT r5 #2;
  // profiles/r06_op_accuracy_*.json); 64-blocks sit at ~0.7x of it.
  // CANON_SMALLM: the no-LDS small-M kernel with a split-K form fixed by (N, K) alone (the lock-step MT decode rows; two
  // interleaved chains per wave, flushed every 64 k like the above).
  // 0: the launcher takes the calling thread's CanonScope mode (none outside a scope = fastest kernel for the shape).
  int canon = 0;
};
constexpr int CANON_NONE = 0, CANON_SEQ = 1, CANON_SMALLM = 2;
constexpr int CANON_KBLOCK = 64;      // CANON_SEQ since round 6: the chain is cut every 64 k of the walk, block sums added in ascending order
// RAII: launches of the calling thread whose GemmArgs::canon is 0 take `mode` until the scope ends (the ss_batch_* entry points
// open one: a packed utterance gets the arithmetic it would get alone or in any other pack)
struct CanonScope {
  explicit CanonScope(int mode);
  ~CanonScope();
  int prev;
};
int canon_mode();                      // the calling thread's current mode
void canon_debug_set(int mode);        // test hook: the calling thread's mode outside any scope (ss_debug_canon)

// Optional per-launch timing with HIP events recorded on the launch stream (bench.py roofline
// leg).  Tile-config classes: see kTileNames in gemm.hip.
constexpr int kNumTileCfg = 32;
void prof_enable(int cls_mask);   // bit i set -> bracket launches of tile config i with events; 0 = off
void prof_reset();
int prof_read(int cls, double* ms_total, double* flops_total, long long* launches, double* bytes_total = nullptr);  // synchronises
const char* prof_cfg_name(int cls);
int prof_totals(int cls, double* flops, double* bytes, long long* launches);   // always-on census since library load
int prof_read_issued(int cls, double* issued_flops_total);   // of the bracketed launches: FLOPs the kernel ISSUES as MFMAs (the Winograd forms issue 4 ceil(k/3) / (2 k) of the algorithmic count)
// Per-shape dispatch table (which kernel class took which (N, taps, Cin, operands) shape): SS_SHAPE_LOG=<path> writes it at exit;
// prof_shape_log(1) collects it from now on in-process, prof_shape_dump writes "class N taps Cin operands launches mean_rows
// gflop_per_launch mbyte_per_launch" lines into buf (returns the bytes needed, buf may be null) -- bench.py's `dispatch` object.
void prof_shape_log(int on);
int prof_shape_dump(char* buf, int cap);

// Event-profiler scope shared by the kernel launchers (gemm.hip, conv_sk.hip).
struct ProfRec { hipEvent_t e0, e1; double flops, bytes, issued; int cls; };
int prof_begin(const GemmArgs& a, hipStream_t stream, int cls, ProfRec& rec, bool& prof);
int prof_end(hipStream_t stream, ProfRec& rec, bool prof);

// Stream-K hand-off state (partial-tile workspace, ticket counter, per-workgroup flags) of ONE execution context: an
// ss_model / ss_vocoder handle owns one and frees it with the handle; every C-ABI entry point opens an SkScope, and the
// stream-K launchers take the scoped workspace of the calling thread (a context is driven by one host thread on one stream
// at a time -- the C ABI's contract -- so nothing here needs a lock).  Launches outside any scope (ss_op_* unit-test entry
// points) fall back to a process-wide table keyed by (device, stream).
struct SkWorkspace {
  int dev = -1, cus = 0;
  float* ws = nullptr;                 // [cus] parked partial tiles of 128 KB (conv_sk2: 256 x 128, conv_sk: 2 x 128 x 128 per CU)
  unsigned* sync1 = nullptr;           // conv_sk: ticket counters, time-out counter, flags
  unsigned* sync2 = nullptr;           // conv_sk2
  unsigned* sync3 = nullptr;           // ffn_fused: per-tile arrival counters (zero between launches)
  unsigned base1[8] = {0, 0, 0, 0, 0, 0, 0, 0}, epoch1 = 0;
  unsigned base2 = 0, epoch2 = 0;
  unsigned long long* dbg = nullptr;   // diagnostic builds only
  CtxDispatch disp;                    // this context's private copy of the dispatch settings (dispatch.hpp); SkScope makes it the thread's disp()
};
SkWorkspace* sk_workspace_new();
void sk_workspace_free(SkWorkspace* w);
struct SkScope {                       // RAII: the calling thread's stream-K launches use `w`, and its launchers read `w`'s dispatch
  explicit SkScope(SkWorkspace* w);    // settings (refreshed here, at the start of an entry point, if the process settings moved), until the scope ends
  ~SkScope();
  SkWorkspace* prev;
  const Dispatch* prev_disp;
};
// the workspace a launch on `stream` should use (scoped one, else the fallback entry), device memory allocated; nullptr + rc on failure
int sk_workspace_acquire(hipStream_t stream, SkWorkspace** out);
int sk_workspace_error_count();        // bounded-wait time-outs recorded by any live workspace (must stay 0)

// Persistent stream-K conv-GEMM (conv_sk.hip): 128 x BN tiles, the (tile, k-step) space is cut into
// equal contiguous ranges, one per resident workgroup, so every CU gets the same MFMA work whatever
// the tile count.  g_force > 0 fixes the grid size (tuning hook).
bool conv_sk_eligible(const GemmArgs& a);
int launch_conv_sk(const GemmArgs& a, hipStream_t stream, int g_force = 0);
int conv_sk_error_count();
void conv_sk_set_groups(int on);   // tuning hook: XCD tile grouping on/off

// Second-generation stream-K kernel for N % 128 == 0 (conv_sk2.hip): 256 x 128 tiles, one workgroup per CU, 3-stage
// LDS-DMA ring, software-pipelined k-loop, wait-free two-contributor hand-off.  g_force > 0 fixes the grid size.
bool conv_sk2_eligible(const GemmArgs& a);
int launch_conv_sk2(const GemmArgs& a, hipStream_t stream, int g_force = 0);
int conv_sk2_error_count();

// Row bound of the slab kernels (conv_slab / conv_pair / resblock_fused / conv_c16 / c32 / c64 / c64w): they address activations as
// base + (size_t)row * ld -- 64-bit -- so a tensor may pass 2^31 BYTES (a 256-utterance pack's 16-channel stage: rounds 1-5 sent such
// tensors to the generic tiles on a byte bound inherited from the buffer-addressed stream-K kernels); rows and segment tables are int.
inline bool slab_rows_ok(long long rows) { return rows > 0 && rows < (1ll << 30); }

// Slab conv for the narrow vocoder stages (conv_slab.hip): C, N in {16, 32}, weights + input slab in LDS.
bool conv_slab_eligible(const GemmArgs& a);
int launch_conv_slab(const GemmArgs& a, hipStream_t stream);

// Fused resblock half y = conv2(lrelu(conv1(lrelu(x)) + b1)) + b2 + x [+ R2] [/ div] for the narrow stages
// (both weight matrices and the intermediate slab in LDS; conv_slab.hip).  A must not alias C.
bool conv_pair_eligible(int C, int taps, int dil, int lda, int ldc, int nseg, long long M);
int launch_conv_pair(const float* A, int lda, const float* W1, const float* b1, const float* W2, const float* b2, float* C,
                     int ldc, const float* R2, int ldr2, float div, float* C2, int ldc2, float c2_slope, int Cch, int taps,
                     int dil, int M, int in_len, float slope, const int* segs, int nseg, hipStream_t stream);

// Fused ResBlock of the narrow stages (resblock.hip): the three (dilated conv, conv, residual) pairs of one ResBlock in one
// persistent launch, raw residual stream in registers, activated / intermediate slabs in LDS, Y = [R2 +] resblock(X) [/ div].
// W1 / B1 / W2 / B2 / dil: the three pairs' matrices ([C][taps*C] tap-major), biases and dilations.  X must not alias Y.
bool resblock_fused_eligible(int C, int taps, const int* dil, int ldx, int ldy, int nseg, long long M);
int launch_resblock_fused(const float* X, int ldx, const float* const* W1, const float* const* B1, const float* const* W2,
                          const float* const* B2, const int* dil, float* Y, int ldy, const float* R2, int ldr2, float div, int C,
                          int taps, int M, float slope, const int* segs, int nseg, hipStream_t stream);

// Slab conv with streamed weights for the 64-channel vocoder stage of a packed batch (conv_c64.hip): Cin = N = 64, "same" rows,
// input leaky-ReLU applied while the slab is staged, W fragments straight from L2.
bool conv_c64_eligible(const GemmArgs& a);
bool conv_c64_enabled();
int launch_conv_c64(const GemmArgs& a, hipStream_t stream);
// conv_c64w.hip: the same launches in Winograd F(2,3) form (needs a.Wwino; k = 11 at dilation 5 stays on conv_c64)
bool conv_c64w_eligible(const GemmArgs& a);       // call with conv_c64_eligible(a) already true
int launch_conv_c64w(const GemmArgs& a, hipStream_t stream);
int launch_wino_pack(const float* W, float* WW, int C, int taps, hipStream_t stream);   // WW: C * ceil(taps/3) * 4 * C floats
void conv_c64w_debug(int enable);                 // A/B: 0 off, 1 on, -1 keep
bool conv_c64w_enabled();
// the same kernel at 128 channels (one workgroup per CU): the 128-channel stage's ResBlock convs instead of conv_sk2<128> + twins
bool conv_c128w_eligible(const GemmArgs& a);
int launch_conv_c128w(const GemmArgs& a, hipStream_t stream);
void conv_c128w_debug(int enable);
bool conv_c128w_enabled();
// ... at 256 channels (two slab phases of 128 input channels, two column halves; one workgroup per CU): the 256-channel stage's ResBlock convs
bool conv_c256w_eligible(const GemmArgs& a);
int launch_conv_c256w(const GemmArgs& a, hipStream_t stream);
void conv_c256w_debug(int enable);
bool conv_c256w_enabled();
// ... and at 32 channels (three workgroups per CU): the per-conv launches of the 32-channel stage (k = 11)
bool conv_c32w_eligible(const GemmArgs& a);       // call with conv_c32_eligible(a) already true
int launch_conv_c32w(const GemmArgs& a, hipStream_t stream);
void conv_c32w_debug(int enable);
bool conv_c32w_enabled();
void conv_c64_debug(int enable);          // A/B: 0 routes the stage back to conv_sk2<64> (pre-activated twins), 1 on, -1 keep

// The same for the 32-channel stage (conv_c32.hip): each ResBlock conv as its own launch instead of one fused launch per ResBlock.
bool conv_c32_eligible(const GemmArgs& a);
bool conv_c32_enabled();
int launch_conv_c32(const GemmArgs& a, hipStream_t stream);
void conv_c32_debug(int enable);

// ... and for the 16-channel stage (conv_c16.hip): the whole 16 x 16k weight matrix in registers.
bool conv_c16_eligible(const GemmArgs& a);
bool conv_c16_enabled();
int launch_conv_c16(const GemmArgs& a, hipStream_t stream);
void conv_c16_debug(int enable);

// Row-tile linear layer for K = 256 projections of packed batches (rtlin.hip): the row tile (LayerNorm-ed when a.ln_g is set) in
// LDS, weight fragments straight from L2 to registers, bias / activation / alpha / residual or GLU epilogue per 16-column unit.
bool rtlin_eligible(const GemmArgs& a);
bool rtlin_shape_ok(const GemmArgs& a);   // what the kernel can compute at all (rtlin_eligible = this + "worth it at this row count")
int launch_rtlin(const GemmArgs& a, hipStream_t stream);
void rtlin_debug(int grid, int enable);   // tests / A-B: fixed workgroup count (0 = heuristic); enable 0 / 1 (-1: keep)
// The same structure for K = 512 ... 8192 (multiples of 256), N % 256 == 0: the row tile goes through LDS one 256-wide k-block at a
// time, block sums are added in ascending order (the CANON_KBLOCK summation: same bits as conv_gemm_kernel<.., BLK = true>).
bool rtlin_kb_shape_ok(const GemmArgs& a);
bool rtlin_kb_eligible(const GemmArgs& a);
int launch_rtlin_kb(const GemmArgs& a, hipStream_t stream);

// Fused Conformer feed-forward module (ffn.hip): Y = X + alpha * (W2 . SiLU(W1 . LayerNorm(X) + b1) + b2), optionally followed by
// LayerNorm(ln2) over the result rows; one persistent launch, hidden activations stay in registers.  D = 256, F % 64 == 0.
// Y may alias X (in place).
bool ffn_fused_eligible(int D, int F, int act, int M, int ldx, int ldy, bool canon = false);
int launch_ffn_fused(const float* X, int ldx, float* Y, int ldy, const float* ln_g, const float* ln_b, const float* W1,
                     const float* b1, const float* W2, const float* b2, float alpha, const float* ln2_g, const float* ln2_b,
                     int M, int D, int F, hipStream_t stream, int canon = 0);
// canon != 0: every row tile is computed whole by ONE workgroup (wave w contracts hidden units [32 w, 32 w + 32) whatever M is),
// so a row's bits do not depend on the row count / grid; the tile height is picked for the fewest rounds over the CUs.
void ffn_fused_debug_grid(int g);    // tests / tuning: fixed workgroup count (0 = heuristic)
void ffn_fused_debug_rows(int wm);   // tests / tuning: 16-row MFMA tiles per wave (1..4: force; 0: process default; < 0: keep)

// True when launch_conv_gemm would route `a` to the decode GEMV (M <= 4 rows; the only form that honours ln_out).
bool gemv_eligible(const GemmArgs& a);
// True when launch_conv_gemm would route `a` to the small-M kernel (the only one with the fused
// LayerNorm prologue).
bool smallm_eligible(const GemmArgs& a);

// Tuning hook: force the tile of the LDS-tiled kernel (bm = 0 restores the heuristic).
void debug_force_tile(int bm, int bn, int ks);
bool debug_tile_forced();

// Launches on `stream`; returns SS_OK / SS_ERR_*.
int launch_conv_gemm(const GemmArgs& a, hipStream_t stream);

}  // namespace ss
