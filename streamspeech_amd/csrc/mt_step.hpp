// Persistent MT decode step (mt_step.hip): argument block and granule layout.
#pragma once
#include "gemm.hpp"

namespace ss {

typedef unsigned long long mt_u64;

constexpr int MT_D = 512, MT_F = 2048, MT_H = 8, MT_DH = 64, MT_L = 4, MT_SPLITS = 4;
constexpr int MT_PARTV = 66;                                   // m, l, acc[64] of one (head, key range)
constexpr int MT_PART = MT_H * MT_SPLITS * MT_PARTV;           // 2112
// granule offsets inside a layer's region
constexpr int MG_QKV = 0, MG_ATT = MG_QKV + 3 * MT_D, MG_X1 = MG_ATT + MT_D, MG_Q2 = MG_X1 + MT_D, MG_PART = MG_Q2 + MT_D,
              MG_X2 = MG_PART + MT_PART, MG_HID = MG_X2 + MT_D, MG_X = MG_HID + MT_F, MG_LAYER = MG_X + MT_D;
constexpr int MG_ARG = MT_L * MG_LAYER;                        // 2 granules per workgroup: value bits, index
constexpr int MT_MAXG = 256;
constexpr int MG_TOTAL = MG_ARG + 2 * MT_MAXG;
constexpr unsigned long long MT_WAIT_TICKS = 1000000ull;  // bounded wait of an exchange: 10 ms of the 100 MHz wall clock (a healthy exchange takes 2-4 us;
                                                         // a workgroup kept off the chip by another stream's full-chip kernel arrives within ~1 ms)

inline size_t mt_step_granule_bytes() { return (size_t)MG_TOTAL * sizeof(mt_u64); }

struct MtLayerW {
  const float *ln1_g, *ln1_b, *wqkv, *bqkv, *wo, *bo;
  const float *ln2_g, *ln2_b, *wcq, *bcq, *wco, *bco;
  const float *ln3_g, *ln3_b, *w1, *b1, *w2, *b2;
  float* selfbuf;          // KV cache [max_tgt_pos][3 * 512] (q | k | v rows)
  const float* cross;      // [Tp][2 * 512] (k | v), projected by ss_mt_begin
};
struct MtStepArgs {
  MtLayerW L[MT_L];
  const float *lnf_g, *lnf_b, *emb, *pos_table;
  const int* tok;          // the token fed at position pos0 (device; written by the previous step / launch)
  float* feats;            // [n_steps][512] out: LN_f(x), the decoder state of every fed position
  int* next;               // [n_steps] out: the token decided at position pos0 + it (-1: a bounded wait timed out)
  mt_u64* gran;
  unsigned* err;
  unsigned epoch;
  int Tp, pos0, V, pad, eos;
  int n_steps;             // decode steps of this launch (the loop ends early at </s>)
  int search = 0;          // 1 (ss_mt_greedy after its prefix pass): a FED </s> at position > 0 means the search is already over -- decode nothing.
                           // 0 (ss_mt_append's single step): compute what is fed, like the launch-per-op form of the same call
  int min_len, max_len;    // </s> is banned at positions < min_len and forced at positions >= max_len
  float emb_scale;
};


// Up to a.n_steps decode steps on G resident workgroups (64, 128 or 256); outputs a.feats rows and a.next tokens.
int launch_mt_step(const MtStepArgs& a, int G, hipStream_t stream);

}  // namespace ss
