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
  const int* tok;          // the token fed at this position (device; written by the previous step)
  float* feats;            // [512] out: LN_f(x), the decoder state of this position
  int* next;               // out: the next token
  mt_u64* gran;
  unsigned* err;
  unsigned epoch;
  int Tp, pos0, V, pad, eos, ban_eos, force_eos;
  float emb_scale;
};


// One decode step on G resident workgroups (64, 128 or 256); the step's outputs are a.feats and a.next.
int launch_mt_step(const MtStepArgs& a, int G, hipStream_t stream);

}  // namespace ss
