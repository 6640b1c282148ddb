// Multi-head attention kernels (head_dim = 64) for the S2ST path.
//  * rel-pos (Transformer-XL style) chunk-masked self-attention of the Conformer encoder:
//    reference researches/uni_unity/modules/espnet_multihead_attention.py:154-209,
//    score[i,j] = ((q_i+u).k_j + (q_i+v).p[j-i+T-1]) / sqrt(64), key j hidden iff j >= (i/c+1)*c
//    (chunk_unity/models/s2t_conformer.py:195-213).
//  * fairseq MHA (MT decoder, T2U encoder, unit decoder): reference
//    researches/ctc_unity/modules/multihead_attention.py:544-760, q pre-scaled by d_h^-0.5,
//    optional causal triu(-inf, 1) mask, softmax in fp32.
#pragma once
#include "common.hpp"
#include "dispatch.hpp"

namespace ss {

struct AttnArgs {
  const float* Q = nullptr; const float* K = nullptr; const float* V = nullptr; float* O = nullptr;
  int ldq = 0, ldk = 0, ldv = 0, ldo = 0;  // row strides in floats; head h occupies cols [64h, 64h+64)
  int Tq = 0, Tk = 0, H = 0;
  float scale = 1.0f;        // scores *= scale
  int causal = 0;            // key j visible iff j <= i + (Tk - Tq)
  int chunk = 0;             // > 0: key j visible iff j < (i/chunk + 1)*chunk
  int q0 = 0;                // single-utterance rel-pos/chunk form only: query row i is absolute position q0 + i
                             // (keys are 0..Tk-1); used by the incremental streaming encoder (tail queries over all keys)
  int k_mask_tail = 0;       // the last k_mask_tail keys are padding (fairseq key_padding_mask on trailing <pad>)
  // rel-pos extras (null => plain attention); requires q0 + Tq == Tk
  const float* P = nullptr; int ldp = 0;   // projected positional table [2*Tk-1, H*64]
  const float* bias_u = nullptr; const float* bias_v = nullptr;  // [H*64]
  // Ragged batch (nseg > 0): independent utterances packed along the row axis.  segs[4*s] =
  // {q_start, q_len, k_start, k_len}: queries/outputs are rows q_start.. of Q/O, keys rows k_start..
  // of K/V; Tq/Tk above are ignored except max_q (grid sizing).  For the rel-pos form P must point
  // at row 0 of the FULL table (relative offset p_tmax-1) and the kernel slices it per segment.
  const int* segs = nullptr; int nseg = 0; int max_q = 0; int p_tmax = 0;
  // Key-split scratch of the calling context (rel-pos form, single utterance only; nullptr = never split): few query tiles
  // over many keys -- one utterance, or the tail rows of the incremental streaming encoder -- would otherwise run as
  // (query tiles x heads) workgroups that each walk ALL key tiles one after the other.  See attention_relpos_mfma_kernel.
  float* part = nullptr;       // [part_slots][ATTN_PART_FLOATS] parked partial (o, m, l) of a (query tile, head, key split)
  unsigned* cnt = nullptr;     // [cnt_slots] arrivals per (query tile, head); zero between launches
  int part_slots = 0, cnt_slots = 0;
  int ksplit = 0, ktiles_per_split = 0;   // filled in by launch_attention
  // Kernel choice for plain (no rel-pos) attention.  0: by query count (the decode kernel up to 8 queries); 1: always the MFMA tile
  // kernel -- ragged batches whose max_q depends on the pack (T2U encoder rows) must not change kernels with it (pack-invariant bits);
  // the lock-step MT decode (max_q = 1 by construction) keeps 0.
  int no_decode_kernel = 0;
};
constexpr int ATTN_PART_FLOATS = 5 * 256 * 4;   // 5 b128 per thread: o[4], {m, l, -, -}
constexpr int ATTN_PART_SLOTS = 512, ATTN_CNT_SLOTS = 512;
inline size_t attention_split_bytes() { return (size_t)ATTN_PART_SLOTS * ATTN_PART_FLOATS * sizeof(float) + ATTN_CNT_SLOTS * sizeof(unsigned); }
// points a.part / a.cnt into a scratch block of attention_split_bytes() bytes whose counter part is zero
inline void attention_bind_split(AttnArgs& a, void* scratch) {
  a.part = static_cast<float*>(scratch);
  a.cnt = reinterpret_cast<unsigned*>(a.part + (size_t)ATTN_PART_SLOTS * ATTN_PART_FLOATS);
  a.part_slots = ATTN_PART_SLOTS; a.cnt_slots = ATTN_CNT_SLOTS;
}
void attention_debug_q16(int v);       // test hook: 0 = never the few-queries rel-pos form (attention_relpos_q16_kernel), 1 = default
void attention_debug_split(int v);     // test hook: -1 never split, 0 heuristic, n > 0 key tiles per split = n

int launch_attention(const AttnArgs& a, hipStream_t stream);
void attention_debug_no_mfma(int v);   // test hook: 1 routes plain attention to the VALU kernel

}  // namespace ss
