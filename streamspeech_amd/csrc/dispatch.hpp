// Dispatch settings: every tuning / A-B knob and test hook that decides WHICH kernel a launch takes.
//
// Through round 4 these were ~30 plain file-scope globals (`static int g_... = getenv(...)`) written by the ss_debug_* test hooks
// and read by every launching thread.  Now:
//   * `Dispatch` is one plain struct; the PROCESS settings (environment defaults + whatever the ss_debug_* hooks edited) live
//     behind a mutex and carry a generation number;
//   * every execution context (ss_model / ss_vocoder: one host thread at a time, by the C ABI's contract) owns a private COPY,
//     refreshed only at the start of an entry point and only if the generation moved -- never in the middle of a call; the
//     launchers read the calling thread's scoped copy (`disp()`), i.e. memory no other thread writes;
//   * launches outside any context (the ss_op_* unit-test entry points) read a thread-local copy kept the same way.
// Environment variables are read ONCE, when the process settings are first touched.
#pragma once
#include <functional>

namespace ss {

struct Dispatch {
  // attention.hip
  int attn_no_mfma = 0;            // test hook: plain attention on the VALU kernel
  int attn_q16 = 1;                // SS_ATTN_Q16=0: rel-pos attention of <= 48 query rows on the 64-query tile kernel instead of attention_relpos_q16_kernel
  int attn_split = 0;              // -1 never split keys, 0 heuristic, n > 0 key tiles per split        SS_ATTN_NO_SPLIT
  // per-conv slab kernels of the vocoder stages (conv_c16 / c32 / c64.hip) and their Winograd forms (conv_c64w.hip)
  int c16_off = 0, c32_off = 0, c64_off = 0;                                          // SS_NO_CONV_C16 / C32 / C64
  long long c16_min_rows = 131072, c32_min_rows = 65536, c64_min_rows = 32768;        // SS_CONV_C16 / C32 / C64_MIN_ROWS
  int c64w_on = 1, c128w_on = 1, c256w_on = 1, c32w_on = 1;                           // SS_CONV_C64 / C128 / C256 / C32_WINOGRAD
  int c64w_min_k = 3;                                                                 // SS_CONV_C64_WINOGRAD_MIN_K
  long long c128w_min_rows = 65536, c256w_min_rows = 32768;                           // SS_CONV_C128 / C256_MIN_ROWS
  // conv_sk.hip
  int sk_groups = 0;               // XCD tile grouping of the first-generation stream-K kernel (tuning hook)
  // ffn.hip / encoder FFN routing
  int ffn_force_g = 0;             // fixed grid (tests)
  int ffn_wm = 3, ffn_wm_forced = 0;   // 16-row MFMA tiles per wave; forced by SS_FFN_WM / ss_debug_ffn
  int ffn_fusion = 1;              // SS_NO_FFN_FUSION
  int ffn_min_rows = 1000;         // SS_FFN_MIN_ROWS
  // gemm.hip
  int force_bm = 0, force_bn = 0, force_ks = 0;   // ss_debug_force_tile
  double sk_min_flops = 4e9;       // SS_SK_MIN_GFLOP
  // vocoder ResBlock fusion (model.hip)
  int no_resblock_fusion = 0, no_pair_fusion = 0;   // SS_NO_RESBLOCK_FUSION / SS_NO_PAIR_FUSION, ss_debug_force_tile(6 | 3)
  // rtlin.hip
  int rt_off = 0, rt_min_rows = 193, rt_force_g = 0;   // SS_NO_RTLIN, SS_RTLIN_MIN_ROWS, ss_debug_rtlin
  long long rt_min_units = 4000;                       // SS_RTLIN_MIN_UNITS
  long long rt_kb_min_units = 256;                     // SS_RTLIN_KB_MIN_UNITS: (48-row tile, 64-column group) units from which rt_linear_kb takes a K > 256 linear
  int rt_kb_xmap = 1;                                  // SS_RTLIN_KB_XMAP: 0 = every workgroup walks a contiguous (tile, column group) range; 1 = the column groups of a row tile run side by side on one XCD
  int rt_kb_uw = 0;                                    // SS_RTLIN_KB_UW: 16-column units per wave and column group (0: by the launch's unit count; 1 / 2 / 4 force)
};

// The calling thread's settings: the scoped context's private copy, else this thread's own copy of the process settings.
const Dispatch& disp();
// ss_debug_* hooks: edit the process settings (under the lock; bumps the generation -- contexts pick the edit up at their next call)
void dispatch_edit(const std::function<void(Dispatch&)>& fn);

// A context's private copy (member of SkWorkspace: every ss_model / ss_vocoder has one).
struct CtxDispatch {
  Dispatch d;
  unsigned gen = 0xffffffffu;
  const Dispatch* refresh();       // re-copy iff the process generation moved; returns &d
};
struct DispatchScope {             // RAII: disp() of the calling thread = *d until the scope ends
  explicit DispatchScope(const Dispatch* d);
  ~DispatchScope();
  const Dispatch* prev;
};

}  // namespace ss
