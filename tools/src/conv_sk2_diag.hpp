// Diagnostic-only definitions of conv_sk2.hip's hook points (NOT part of the product build: included only when the library
// is built with -DK2_DIAGNOSTIC_BUILD by tools/sk2_bench.py / tools/sk2_timing.py).
//
// K2_ABL=<mask>: timing-only ablation builds -- the kernel's results are WRONG BY DESIGN, only its duration is read:
//   1 no LDS-DMA in the k-loop, 2 no wait + barrier in the k-loop, 4 no ds_reads in the k-loop, 8 no epilogue / hand-off,
//   16 no MFMAs, 32 no hand-off (every part runs the epilogue), 64 no C / C2 stores, 128 no R / R2 loads, 256 one A piece
//   per k-step instead of eight.   (profiles/r02_sk2_ablation*.txt, profiles/r02_sk2_timing_ablation.txt)
// K2_TIMING=1: thread 0 of every workgroup accumulates s_memtime cycles per phase into q.dbg, read back through
//   ss_debug_sk2_timing (profiles/r02_sk2_timing.txt); costs ~6 % of the kernel's cycles.
#pragma once
#ifndef K2_ABL
#define K2_ABL 0
#endif
#ifndef K2_TIMING
#define K2_TIMING 0
#endif
#define K2D_SKIP_DMA(in_loop) (((K2_ABL) & 1) && (in_loop))
#define K2D_SKIP_A_PIECE(in_loop, j) (((K2_ABL) & 256) && (in_loop) && (j) > 0)
#define K2D_SKIP_WAIT_BARRIER (((K2_ABL) & 2) != 0)
#define K2D_SKIP_DS_READ(in_loop) (((K2_ABL) & 4) && (in_loop))
#define K2D_SKIP_TAIL (((K2_ABL) & 8) != 0)
#define K2D_SKIP_MFMA (((K2_ABL) & 16) != 0)
#define K2D_SKIP_HANDOFF (((K2_ABL) & 32) != 0)
#define K2D_SKIP_STORES (((K2_ABL) & 64) != 0)
#define K2D_SKIP_R_LOADS (((K2_ABL) & 128) != 0)
#define K2D_ONE_A_PIECE ((K2_ABL) & 256)
#define K2D_TIMING K2_TIMING
#if K2_TIMING
#define K2D_TIMING_DECL                                                                                              \
  unsigned long long tph[6] = {0, 0, 0, 0, 0, 0}; /* prologue, part set-up, k-loop, hand-off, epilogue, k-steps */   \
  const unsigned long long t_begin = __builtin_readcyclecounter();                                                   \
  unsigned long long tk_last = t_begin;                                                                              \
  auto stamp = [&](int slot) { const unsigned long long now = __builtin_readcyclecounter(); tph[slot] += now - tk_last; tk_last = now; };
#define K2_STAMP(slot) stamp(slot)
#define K2D_COUNT_STEPS(n) tph[5] += (unsigned long long)(n)
#define K2D_TIMING_FLUSH                                                     \
  do {                                                                       \
    if (t == 0 && q.dbg) {                                                   \
      unsigned long long* d = q.dbg + (size_t)w * 8;                         \
      for (int i = 0; i < 6; ++i) d[i] = tph[i];                             \
      d[6] = t_begin; d[7] = __builtin_readcyclecounter();                   \
    }                                                                        \
  } while (0)
#else
#define K2D_TIMING_DECL
#define K2_STAMP(slot) do { } while (0)
#define K2D_COUNT_STEPS(n) do { } while (0)
#define K2D_TIMING_FLUSH do { } while (0)
#endif
