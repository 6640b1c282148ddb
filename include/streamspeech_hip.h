/* streamspeech_hip.h -- C ABI of the MI355X-native StreamSpeech S2ST forward pass.
 *
 * Drop-in boundary: the reference has no FFI -- its S2ST path is torch modules
 * called from the SimulEval agent -- so each entry point below replaces one module call the agent
 * makes (reference agent/speech_to_speech.streamspeech.agent.py, cited per function).  Signatures
 * carry only plain pointers and sizes: `d_*` arguments are DEVICE (HBM) pointers, `h_*` are host
 * pointers, `stream` is a hipStream_t passed as void*.  Every call is stream-ordered unless it
 * says "synchronises".  All arithmetic is FP32.  Return value: 0 = ok, non-zero = SS_ERR_*.
 *
 * Weight ownership: the caller owns one packed FP32 weight blob in HBM (built once from a fairseq
 * state dict by streamspeech_amd/weights.py) and lends it to ss_model_create(); the library keeps
 * borrowed pointers into it; scratch / caches live in scratch sets (ss_scratch_*, hipMalloc).
 */
#ifndef STREAMSPEECH_HIP_H
#define STREAMSPEECH_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SS_ABI_VERSION 2              /* 2 (round 6): scratch sets are objects of their own (ss_scratch_*), SS_ERR_SCRATCH_CAP */

typedef struct ss_model ss_model;      /* StreamSpeechModel replacement (encoder + CTC + MT + T2U + unit decoder): WEIGHTS of one language */
typedef struct ss_vocoder ss_vocoder;  /* CodeHiFiGANVocoderWithDur replacement: WEIGHTS */
typedef struct ss_scratch ss_scratch;  /* everything a call mutates: activations, KV caches, stream-K hand-off state, streaming state */

/* Architecture hyper-parameters (reference researches/ctc_unity/models/streamspeech_model.py:418-430
 * and train_scripts/train.offline-s2st.sh). */
typedef struct ss_config {
  int32_t input_feat, conv_channels, conv_kernel;
  int32_t enc_dim, enc_ffn, enc_heads, enc_layers, dw_kernel;
  int32_t src_vocab, tgt_vocab;
  int32_t mt_layers, dec_dim, dec_ffn, dec_heads;
  int32_t t2u_layers, unit_layers, unit_vocab, ctc_upsample;
  int32_t pad, eos, unk;
  int32_t max_rel_pos;     /* Tmax: the projected rel-pos table covers offsets -(Tmax-1)..Tmax-1 */
  int32_t max_tgt_pos;     /* rows in the MT sinusoid table */
} ss_config;

typedef struct ss_vocoder_config {
  int32_t num_embeddings, embedding_dim, model_in_dim, upsample_initial_channel;
  int32_t n_up;  int32_t upsample_rates[8];  int32_t upsample_kernel_sizes[8];
  int32_t n_res; int32_t resblock_kernel_sizes[4]; int32_t resblock_dilations[4][3];
  int32_t dur_hidden, dur_kernel;
} ss_vocoder_config;

int ss_abi_version(void);
const char* ss_error_string(int code);

/* ---- model lifetime ------------------------------------------------------------------------
 * names[i] / offsets[i] / numels[i]: slot i of the packed blob starts offsets[i] floats into
 * d_blob and holds numels[i] floats.  Replaces load_model_ensemble + model.cuda()
 * (agent :355-420).  Synchronises (builds the per-layer projected rel-pos tables). */
int ss_model_create(const ss_config* cfg, const float* d_blob, size_t blob_floats,
                    const char* const* names, const int64_t* offsets, const int64_t* numels,
                    int n_slots, ss_model** out);
void ss_model_destroy(ss_model* m);

/* ---- scratch sets -----------------------------------------------------------------------------
 * The reference loads one model per language directory (configs/{fr,es,de}-en/, agent :357-401) and PyTorch's caching allocator
 * holds the activations of whatever runs.  Here a weight handle (ss_model / ss_vocoder: borrowed pointers into the blob -- a few KB;
 * the projected rel-pos table and the Winograd weight forms are shared per blob) is separate from a SCRATCH SET, which owns
 * everything a call mutates.  A handle runs on the scratch set it is bound to: its own (made by ss_*_create, empty until used)
 * or, after ss_model_bind_scratch / ss_vocoder_bind_scratch, a shared one -- so L languages on S concurrent streams need S scratch
 * sets and L x S handles, not L x S scratch sets.  Rules: a scratch set (and every handle bound to it) is driven by ONE host
 * thread at a time; a stateful sequence (ss_mt_begin ... ss_mt_append / ss_mt_truncate, ss_encoder_stream_*) stays on one (handle,
 * scratch) pair from start to end; between sequences any handle bound to the set may use it.  Ref-counted: the memory goes when
 * ss_scratch_destroy has been called AND every handle bound to the set is destroyed or re-bound.
 * Buffers grow on demand and never shrink by themselves: ss_scratch_set_cap bounds their sum (a call that would pass it returns
 * SS_ERR_SCRATCH_CAP and leaves the set as it was; 0 = no cap), ss_scratch_trim synchronises the device and lets the largest
 * re-sizable buffers go until at most keep_bytes are held (fixed pieces -- MT cache, token chain, zero-initialised counters -- stay),
 * ss_scratch_bytes reports what is held.  (The stream-K hand-off workspace, 32 MB + flags per set, is not counted.) */
int ss_scratch_create(ss_scratch** out);
void ss_scratch_destroy(ss_scratch* sc);
int ss_scratch_set_cap(ss_scratch* sc, size_t max_bytes);
int ss_scratch_trim(ss_scratch* sc, size_t keep_bytes);
size_t ss_scratch_bytes(ss_scratch* sc);
int ss_model_bind_scratch(ss_model* m, ss_scratch* sc);
int ss_vocoder_bind_scratch(ss_vocoder* v, ss_scratch* sc);

/* ---- a1: OnlineFeatureExtractor.__call__ (agent :66-98) on 16 kHz PCM already in HBM.
 * d_feat must hold ss_fbank_num_frames(n_samples)*80 floats. */
int ss_fbank_num_frames(int n_samples);
int ss_fbank_cmvn(ss_model* m, void* stream, const float* d_pcm16k, int n_samples, float pcm_scale,
                  float* d_feat, int* h_n_frames);

/* Waveform front-end: rational polyphase resampling on the device, standing in for
 * the sox `rate` effect of convert_waveform (fairseq/data/audio/audio_utils.py:53-62; third-party
 * arithmetic outside the fbank-input parity contract).  d_out[k] = sum_m d_in[m] *
 * d_taps[half_len + k*down - m*up] for k < n_out; n_out is normally ceil(n_in*up/down), up/down in
 * lowest terms, d_taps [2*half_len+1] is the host-designed low-pass with gain `up`
 * (streamspeech_amd/frontend.py design_filter). */
int ss_resample(void* stream, const float* d_in, int64_t n_in, int up, int down, const float* d_taps,
                int half_len, float* d_out, int64_t n_out);

/* Offline driver only: d_out[r] = max over the vocabulary, ids mask0..2 skipped (< 0: none), of
 * log_softmax(d_logits[r, :]) -- the per-position score `lprobs.max(dim=2)` of the reference's offline unit search
 * (researches/ctc_unity/ctc_generator.py:55-63: pad / unk / eos set to -inf AFTER the softmax), which fairseq-generate
 * prints as `H-`/`D-` score (their sum) and `P-` positional scores (fairseq/fairseq_cli/generate.py:274-291).
 * d_logits [rows, vocab] dense, as ss_t2u_units returns them. */
int ss_row_max_logprob(void* stream, const float* d_logits, int rows, int vocab, int mask0, int mask1, int mask2,
                       float* d_out);

/* Normalised (log-)probabilities of dense logit rows [rows, vocab] -> d_out [rows, vocab]: log_softmax (as_probs = 0) or
 * softmax (as_probs = 1) over the vocabulary, then ids mask0 / mask1 (< 0: none) set to -inf (0 for probabilities) -- what
 * the reference forms with `model.get_normalized_probs` (researches/ctc_unity/models/streamspeech_model.py via
 * fairseq_model.py:60-77) and `lprobs[:, :, pad] = lprobs[:, :, unk] = -inf` (agent/ctc_decoder.py:52-60) when a caller
 * asks the CTC decoder for `lprobs`.  Off the timed path (the greedy searches never form log-probabilities). */
int ss_log_softmax(void* stream, const float* d_logits, int rows, int vocab, int mask0, int mask1, int as_probs,
                   float* d_out);

/* ---- a2-a7: model.encoder(src_tokens, src_lengths) for one utterance (agent :433-435 ->
 * chunk_unity/models/s2t_conformer.py:111-163).  d_fbank [T,80] -> d_enc_out [T',256].
 * attn_chunk = encoder.chunk_size, conv_chunk = ChunkCausalConv1d.chunk_size as the agent sets
 * them (:395-413); values >= 999 mean "offline" (no chunking). */
int ss_encoder_out_len(int T);
int ss_encoder_forward(ss_model* m, void* stream, const float* d_fbank, int T, int attn_chunk,
                       int conv_chunk, float* d_enc_out);

/* Incremental twin for streaming (replaces the full recompute of the encoder at
 * every policy() call, agent :425-435): same inputs and output as ss_encoder_forward for the
 * fbank of ALL audio received so far, but rows that were final at the previous call (nothing they
 * can see -- chunk attention, chunk-causal convs -- can still change) are served from the handle's
 * cache and only the remaining rows go through the conformer layers.  *n_final = rows final after
 * this call, *n_computed = rows recomputed by this call (either may be NULL).  The cache belongs to
 * the handle: one utterance at a time, ss_encoder_stream_reset between utterances (the agent's
 * reset()); a change of chunk sizes or a shorter input resets it implicitly. */
int ss_encoder_stream_reset(ss_model* m);
/* Trailing fbank frames whose values may still change when more audio arrives (default 0).  The agent's
 * front-end resamples the WHOLE sample history at every call (agent :86-89, convert_waveform); a zero-padded
 * FIR recomputes its last few output samples once the future exists, so with a non-16 kHz source the newest
 * fbank frame is not settled and rows whose cone reaches it must not be cached as final: set 1. */
int ss_encoder_stream_set_tail(ss_model* m, int unsettled_fbank_frames);
int ss_encoder_stream_forward(ss_model* m, void* stream, const float* d_fbank, int T, int attn_chunk,
                              int conv_chunk, float* d_enc_out, int32_t* n_final, int32_t* n_computed);
/* Deferred time-out check (off by default).  On a scratch set that runs the persistent forms (ss_mt_set_persistent), the call above ends
 * with a stream synchronisation: it must know that none of its persistent layer launches timed out before anybody reads d_enc_out.
 * A caller that queues more work behind the encoder anyway and synchronises once for all of it -- the agent's policy(): both CTC heads,
 * then one device-to-host copy -- switches the check to deferred (on = 1): ss_encoder_stream_forward then returns without waiting, and
 * ss_encoder_stream_status waits for the stream, sets *repeat = 1 if a launch timed out (the scratch set has then left the persistent
 * form and the failed call's rows are not final any more) and 0 otherwise.  With *repeat = 1 everything computed from d_enc_out since
 * the forward call is void: call ss_encoder_stream_forward again (it now runs one launch per op and waits) and recompute.  Until the
 * status call returns 0, d_enc_out must not be trusted.  A forward call with a check still outstanding settles it first. */
int ss_encoder_stream_set_deferred(ss_model* m, int on);
int ss_encoder_stream_status(ss_model* m, void* stream, int32_t* repeat);

/* ---- a8: CTCDecoder.generate (agent/ctc_decoder.py:39-111): head 0 = source_unigram (ASR),
 * 1 = ctc_target_unigram (ST).  Outputs (device int32): raw argmax per frame [Tp], collapsed
 * tokens / frame index [<=Tp], *d_count.  d_logits may be NULL (else [Tp,vocab] is written). */
int ss_ctc_greedy(ss_model* m, void* stream, int head, const float* d_enc_out, int Tp,
                  int32_t* d_raw, int32_t* d_tokens, int32_t* d_index, int32_t* d_count,
                  float* d_logits);

/* Decode-step form of this context (the n = 1 calls of ss_mt_append / the loop inside ss_mt_greedy; reference: one
 * EnsembleModel.forward_decoder call of agent/sequence_generator.py:592-673 per generated token).  workgroups = 0 (default, or
 * the SS_MT_PERSISTENT environment variable at context creation): one launch per op, 35 dependent kernels per token.
 * workgroups = 64 | 128 | 256: the whole step as ONE persistent launch (csrc/mt_step.hip: phases exchange their output vectors
 * through agent-scope {epoch, value} granules; every wait is bounded and counted by ss_debug_sk_errors).  All workgroups of a
 * launch must become resident: meant for a context that decodes one utterance at a time on an otherwise lightly loaded
 * device (the SimulEval agent), at most 8 such contexts concurrently at 64 workgroups.
 * The setting belongs to the SCRATCH SET the handle is bound to when this is called (the granule region lives there): a handle bound
 * to another set afterwards runs with that set's setting. */
int ss_mt_set_persistent(ss_model* m, int workgroups);
/* The current setting.  It drops to 0 by itself when a launch of the persistent step times out: the kernel then publishes -1
 * as the next token, ss_mt_greedy (which reads the token chain back anyway) repeats the whole search with one launch per op and
 * says so on stderr; callers of ss_mt_append that read *d_next < 0 do the same (engine.HipModel.mt_append). */
int ss_mt_get_persistent(ss_model* m);

/* ---- a9-a10: first-pass MT decoder with KV cache (the reference re-runs it on the whole prefix
 * every step, agent/sequence_generator.py:313-346; per-position results are identical).
 * ss_mt_begin: bind encoder output, project cross-attention K/V for all layers, reset the cache.
 * ss_mt_append: feed n tokens occupying positions pos0..pos0+n-1 (position 0 is the leading
 *   </s>); writes features (post final LayerNorm, = mt_decoder(..., features_only=True)) for
 *   those positions to d_feats [n,512] if non-NULL, and the greedy next token after the LAST fed
 *   position to *d_next: argmax over the vocabulary with `pad` never selected, `eos` banned when
 *   ban_eos, and forced to eos when force_eos (agent/sequence_generator.py:350-372 at beam 1). */
int ss_mt_begin(ss_model* m, void* stream, const float* d_enc_out, int Tp);
int ss_mt_append(ss_model* m, void* stream, const int32_t* d_tokens, int n, int pos0, int ban_eos,
                 int force_eos, float* d_feats, int32_t* d_next, int n_tail_pad);
/*   n_tail_pad: the last n_tail_pad fed tokens are <pad> (whole-word mode feeds one, agent :576-584):
 *   they take the zero positional row and are masked as keys (fairseq self_attn_padding_mask). */
/* The whole beam-1 search of SequenceGenerator.generate_decoder (agent/sequence_generator.py:165-582)
 * in one call: begin + prefix pass + autoregressive steps until </s> or max_len (forced </s>), with
 * the token chain kept on the device.  h_prefix [n_prefix] host ids (no leading </s>).
 * h_out_tokens (host, >= max_len+1-n_prefix ints) receives the tokens generated after the prefix,
 * including the final </s>; d_feats [max_len+2, dec_dim] receives the decoder states of every fed
 * position; *h_n_feats = number of valid rows = 1 + n_prefix + (*h_n_out - 1).  Synchronises. */
int ss_mt_greedy(ss_model* m, void* stream, const float* d_enc_out, int Tp, const int32_t* h_prefix,
                 int n_prefix, int max_len, int min_len, int32_t* h_out_tokens, int* h_n_out,
                 float* d_feats, int* h_n_feats);
/* Truncate the self-attention cache to `len` positions (whole-word rollback, agent :540-574). */
int ss_mt_truncate(ss_model* m, int len);

/* ---- a11-a13: synthesizer_encoder + CTCTransformerUnitDecoder + CTC unit search
 * (agent :661-717).  d_mt_feats [n,512] -> raw argmax [25n], collapsed unit-vocabulary tokens
 * (blank 1004 / pad dropped) and *d_count; d_logits optional [25n,1005].  t2u_causal = the
 * checkpoint's --uni-encoder flag; mask_eos = the offline generator's extra eos mask
 * (researches/ctc_unity/ctc_generator.py:58). */
int ss_t2u_units(ss_model* m, void* stream, const float* d_mt_feats, int n, int t2u_causal,
                 int mask_eos, int32_t* d_raw, int32_t* d_tokens, int32_t* d_count, float* d_logits,
                 int n_tail_pad);
/*   n_tail_pad: trailing rows of d_mt_feats that belong to <pad> tokens: masked as keys in the T2U
 *   encoder, the unit decoder's self-attention (25 positions each) and its cross-attention, but
 *   still decoded (the reference emits their units too, agent :661-717). */

/* ---- a14-a15: CodeHiFiGANVocoderWithDur.forward (agent/tts/vocoder.py:48-60). -------------- */
/* d_blob must be device-visible when this is called (the upload complete, not merely queued on some stream): the Winograd forms
 * of the ResBlock conv weights are made here on the null stream (once per blob and device -- later contexts over the same blob
 * borrow them) and the call returns with the device synchronised.  The same holds for ss_model_create (projected rel-pos table). */
int ss_vocoder_create(const ss_vocoder_config* cfg, const float* d_blob, size_t blob_floats,
                      const char* const* names, const int64_t* offsets, const int64_t* numels,
                      int n_slots, ss_vocoder** out);
void ss_vocoder_destroy(ss_vocoder* v);
/* Optional, OFF by default (the default path is exact f32, the reference's arithmetic).  on != 0: the C >= 64 generator
 * convs of this handle (hifigan.py:52-172) contract with three bf16 MFMAs per k-slice on operands split into
 * bf16(x) + bf16(x - bf16(x)) (16 significant bits), f32 accumulation -- waveform within 1e-3 RMS of the f32 path
 * (tests/test_bf16x3_gpu.py), durations untouched (the duration predictor and every argmax stage stay f32). */
int ss_vocoder_set_bf16x3(ss_vocoder* v, int on);
/* d_codes [K] int32 unit ids (0..999).  dur_prediction != 0 runs the duration predictor, else every
 * unit lasts one frame; d_forced_dur (may be NULL) overrides both.  d_wav must hold
 * wav_capacity floats; *h_n_samples = 320 * sum(dur).  d_dur [K] int32.  Synchronises once
 * (the frame count sizes the generator launches). */
int ss_vocoder_forward(ss_vocoder* v, void* stream, const int32_t* d_codes, int K, int dur_prediction,
                       const int32_t* d_forced_dur, float* d_wav, int64_t wav_capacity,
                       int32_t* d_dur, int64_t* h_n_samples);

/* ---- ragged-batch twins (BASELINE.json configs[3]/[4]: many utterances per GPU) --------------
 * B independent utterances packed along the row axis (no padding; each keeps the B = 1 arithmetic
 * of the entry points above -- SURVEY.md H2b).  h_* arrays are host arrays of length B; packed
 * device buffers are the concatenation of the per-utterance arrays in batch order. */
int ss_batch_fbank_cmvn(ss_model* m, void* stream, int B, const float* d_pcm, const int64_t* h_pcm_start,
                        const int32_t* h_n_samples, float pcm_scale, float* d_feat, int32_t* h_T);
int ss_batch_encoder_forward(ss_model* m, void* stream, int B, const float* d_fbank, const int32_t* h_T,
                             int attn_chunk, int conv_chunk, float* d_enc_out, int32_t* h_Tp);
int ss_batch_ctc_greedy(ss_model* m, void* stream, int head, int B, const float* d_enc_out,
                        const int32_t* h_Tp, int32_t* d_raw, int32_t* d_tokens, int32_t* d_index,
                        int32_t* d_counts);
/* Lockstep beam-1 search from [</s>] (offline: no prefix).  h_max_len[b] = forced-</s> step of
 * utterance b.  h_out_tokens [B][out_stride] receives the generated tokens (incl. the final </s>),
 * h_n_out[b] their number (= rows of valid decoder states); d_feats is [B][feat_rows][dec_dim].
 * B <= 256.  Synchronises. */
int ss_batch_mt_greedy(ss_model* m, void* stream, int B, const float* d_enc_out, const int32_t* h_Tp,
                       const int32_t* h_max_len, int min_len, int32_t* h_out_tokens, int out_stride,
                       int32_t* h_n_out, float* d_feats, int feat_rows);
int ss_batch_t2u_units(ss_model* m, void* stream, int B, const float* d_feats, int feat_rows,
                       const int32_t* h_n, int t2u_causal, int mask_eos, int32_t* d_raw, int32_t* d_tokens,
                       int32_t* d_counts);
/* d_codes / d_dur / d_forced_dur packed [sum K]; d_wav packed, utterance b at h_wav_start[b] with
 * h_n_samples[b] samples.  Synchronises once. */
int ss_batch_vocoder_forward(ss_vocoder* v, void* stream, int B, const int32_t* d_codes, const int32_t* h_K,
                             int dur_prediction, const int32_t* d_forced_dur, float* d_wav,
                             int64_t wav_capacity, int32_t* d_dur, int64_t* h_wav_start,
                             int64_t* h_n_samples);

/* ---- profiling hooks for bench.py's roofline leg: bracket every conv-GEMM launch whose tile
 * configuration is in cls_mask with HIP events recorded on the launch stream.  ss_prof_read
 * synchronises on the recorded events and returns the summed kernel time, the summed algorithmic
 * FLOPs (2*M*N*taps*Cin per launch), the launch count and the summed algorithmic bytes (weights,
 * inputs, outputs, residuals once each) of class `cls`. */
int ss_prof_enable(int cls_mask);
int ss_prof_reset(void);
int ss_prof_read(int cls, double* h_ms_total, double* h_flops_total, int64_t* h_launches,
                 double* h_bytes_total);
/* Always-on census since the library was loaded (no events, no synchronisation): launches and algorithmic FLOPs /
 * bytes of tile class `cls` over the WHOLE process -- the denominator for a rocprofv3 kernel-stats or PMC run of the
 * same process (bench.py prints them as "process_census"). */
int ss_prof_totals(int cls, double* h_flops_total, double* h_bytes_total, int64_t* h_launches);
/* Of the launches ss_prof_read covers: the FLOPs the kernels ISSUE as MFMAs.  Equal to the algorithmic count except for the Winograd
 * F(2,3) classes, which issue 4 ceil(k/3) / (2 k) of it -- lets bench.py report a matrix-core busy fraction next to the algorithmic one. */
int ss_prof_read_issued(int cls, double* h_issued_flops_total);
/* Dispatch table: which kernel class took which conv / linear shape.  ss_prof_shape_log(1) starts a fresh in-process collection,
 * ss_prof_shape_dump writes "class N taps Cin operands launches mean_rows gflop_per_launch mbyte_per_launch" lines (operands bit mask:
 * 1 residual, 2 second residual, 4 twin output, 8 input activation, 16 output activation, 32 GLU, 64 ragged segments) into buf and
 * returns the bytes needed (buf may be NULL).  SS_SHAPE_LOG=<path> writes the same table at process exit. */
int ss_prof_shape_log(int on);
int ss_prof_shape_dump(char* buf, int cap);
int ss_prof_num_classes(void);
const char* ss_prof_class_name(int cls);

/* Pack-invariant arithmetic of the ragged-batch twins (default on; SS_PACK_INVARIANT=0 at context creation or on = 0 here switch it
 * off).  The reference decodes ONE utterance per call (agent/speech_to_speech.streamspeech.agent.py:425-478), so an utterance's ids
 * cannot depend on what else is being decoded.  With on = 1 every stage of ss_batch_encoder_forward / ss_batch_ctc_greedy /
 * ss_batch_mt_greedy / ss_batch_t2u_units computes a packed utterance with a summation order that is a function of that utterance
 * alone: every GEMM is one accumulator chain per output element over ascending k (LDS-tiled kernel without split-K, the row-tile
 * kernel, stream-K cut on whole tiles: the same bits), the fused FFN computes every row tile whole in one workgroup, the LayerNorm
 * form and the attention kernel are fixed per layer, the lock-step MT decode uses one split-K form per layer shape -- so the
 * logits of an utterance are bit-identical alone, in any pack and at any position of a pack (tests/test_margin_gpu.py,
 * tests/test_pack_invariance_gpu.py).  on = 0: the fastest kernel for each launch's row count (stream-K k-splits, split-K tiles). */
int ss_model_set_pack_invariant(ss_model* m, int on);
int ss_model_get_pack_invariant(ss_model* m);
/* Test hook: arithmetic mode of the ss_op_* entry points called from the calling thread -- 0 the fastest kernel per shape, 1 the
 * pack-invariant one-chain form, 2 the fixed small-M form of the lock-step decode rows (the model entry points set their own). */
int ss_debug_canon(int mode);

/* Test hook (tests/test_margin_gpu.py): the dense logits [rows, cols] the LAST ss_batch_ctc_greedy / ss_batch_t2u_units call of
 * this context took its arg-max over (they live in the context's scratch until the next call that reuses it).  d_out == NULL:
 * size query only.  Lets a test measure top-1 / top-2 margins of the packed-batch path against the single-utterance path. */
int ss_debug_last_logits(ss_model* m, void* stream, float* d_out, int64_t cap_floats, int* h_rows, int* h_cols);

/* Unit-test entry of the fused Conformer feed-forward kernel (csrc/ffn.hip; what ss_batch_encoder_forward launches twice per
 * layer on packed batches): dY = dX + alpha * (W2 . SiLU(W1 . LayerNorm(dX; ln_g, ln_b) + b1) + b2), then LayerNorm(ln2_g, ln2_b)
 * over the result rows when ln2_g != NULL -- FeedForwardModule.forward + the 0.5-residual wiring and final_layer_norm of
 * ChunkConformerEncoderLayer.forward (researches/chunk_unity/modules/conformer_layer.py:152-164, 254-312).  D must be 256,
 * F % 64 == 0; W1 [F, D], W2 [D, F] row-major; dY may alias dX. */
int ss_op_ffn_fused(void* stream, const float* dX, int ldx, float* dY, int ldy, const float* ln_g, const float* ln_b,
                    const float* dW1, const float* db1, const float* dW2, const float* db2, float alpha,
                    const float* ln2_g, const float* ln2_b, int M, int D, int F);
/* A/B + test hook of the same kernel: grid > 0 fixes its workgroup count (0: heuristic); row_tiles_per_wave 1..4 forces 16- .. 64-row
 * tiles (0: back to the process default: SS_FFN_WM if set, else 48 rows / the pack-invariant form's own choice; -1: keep); enable 0 / 1 switches its use by ss_batch_encoder_forward off / on (-1: keep). */
int ss_debug_ffn(int grid, int row_tiles_per_wave, int enable);
/* A/B + test hook of the row-tile linear kernel (csrc/rtlin.hip: every K = 256 linear of more than 192 rows that goes through
 * ss_op_conv_gemm / the model entry points): grid > 0 fixes its workgroup count (0: heuristic); enable 0 / 1 routes those linears
 * back to the LDS-tiled kernel / to it (-1: keep). */
int ss_debug_rtlin(int grid, int enable);
/* A/B hook of the 64-channel vocoder-stage kernel (csrc/conv_c64.hip): 0 routes that stage's convs back to the stream-K kernel
 * with pre-activated twin tensors (round 3), 1 to it, -1 keeps the setting; 4 / 5 switch the Winograd F(2,3) form of those convs
 * (csrc/conv_c64w.hip) off / on without touching the first setting; 6 / 7 route the 128-channel stage's ResBlock convs to conv_sk2<128>
 * with twins / to the same Winograd kernel at 128 channels; 8 / 9 the same for the 256-channel stage (the kernel at CH = 256: two slab
 * phases of 128 input channels, two column halves). */
int ss_debug_conv_c64(int enable);
/* The same for the 32-channel stage (csrc/conv_c32.hip): 0 = one fused launch per ResBlock (round 3), 1 = one launch per conv;
 * 4 / 5 = the Winograd form of those per-conv launches (csrc/conv_c64w.hip at 32 channels) off / on. */
int ss_debug_conv_c32(int enable);
int ss_debug_conv_c16(int enable);       /* ... and for the 16-channel stage (csrc/conv_c16.hip) */
/* Persistent layer launches of the incremental streaming encoder (csrc/enc_step.hip: two launches + the attention kernel per layer
 * instead of eleven, on a scratch set whose persistent forms are on -- ss_mt_set_persistent -- and for calls with <= 48 rows to compute;
 * SS_NO_ENC_STEP=1 keeps one launch per op): how many such launches this process has made (tests). */
int64_t ss_debug_enc_step_launches(void);
/* Unit-test entry of the LayerNorm-prologue linears (what ln_linear() in model.hip issues for QKV / pointwise conv 1 / the FFNs
 * of one utterance): dC = epi(LayerNorm(dX; ln_g, ln_b, eps 1e-5) . dW^T + dbias), epi as ss_op_conv_gemm (act, alpha, + dR, glu).
 * Served by the small-M kernel (<= 192 rows) or the row-tile kernel (K = 256, more rows); SS_ERR_ARG otherwise. */
int ss_op_ln_linear(void* stream, const float* dX, int ldx, const float* ln_g, const float* ln_b, const float* dW,
                    const float* dbias, const float* dR, int ldr, float* dC, int ldc, int M, int N, int K, int act, float alpha,
                    int glu);

/* Tuning / A-B hook (tools/conv_bench.py, tests): bm = 0 heuristic; 1 route every eligible launch to the first-generation
 * stream-K kernel with a grid of ks workgroups (ks = 0 -> 2 per CU; bn = 8: XCD tile groups); 2 no slab kernel; 3 the
 * narrow-stage resblock pairs of the vocoder as two launches; 4 second-generation stream-K with a grid of ks; 5 its split-bf16
 * form; 6 narrow-stage ResBlocks as separate launches; 32 / 64 / 128 a forced tile of the LDS-tiled kernel (bn, ks = KS*10+PD).
 * Any other bm: SS_ERR_ARG. */
int ss_debug_force_tile(int bm, int bn, int ks);
/* Number of bounded-spin time-outs the stream-K kernel has recorded (any value but 0 is a bug). */
int ss_debug_sk_errors(void);
/* Test hook for the key-split form of the single-utterance rel-pos attention (csrc/attention.hip): -1 never split, 0 the
 * launch heuristic (default), n > 0 force n key tiles (of 64 keys) per split. */
int ss_debug_attention_split(int v);
/* Test hook for the few-queries form of the rel-pos attention (attention_relpos_q16_kernel: <= 48 query rows over all keys, the
 * incremental streaming encoder's calls): 0 = off (the 64-query tile kernel takes those launches), 1 = on (default). */
int ss_debug_attention_q16(int v);
/* Test hook: the next time-out check of this handle's scratch set (ss_encoder_stream_status, or the end of a non-deferred
 * ss_encoder_stream_forward) reports a time-out although none happened -- drives the fall-back and the repeat protocol. */
int ss_debug_enc_step_inject_timeout(ss_model* m);
/* Test hook: the next launch of the persistent MT decode step behaves as if a bounded wait had timed out (it publishes -1),
 * without touching the counter above.  tests/test_mt_persistent_gpu.py drives the fall-back with it. */
int ss_debug_mt_inject_timeout(ss_model* m);

/* ---- op-level entry points (unit tests of single kernels; same launchers the stages use) ---- */
int ss_op_conv_gemm(void* stream, const float* dA, int lda, const float* dW, const float* dbias,
                    const float* dR, int ldr, const float* dR2, int ldr2, float* dC, int ldc,
                    int M, int N, int Cin, int taps, int dil, int stride, int pad, int in_len,
                    int chunk, int in_act, float in_slope, int act, float alpha, float div, int glu);
int ss_op_layernorm(void* stream, const float* dx, int ldx, float* dy, int ldy, const float* dg,
                    const float* db, int M, int D, float eps);
int ss_op_attention(void* stream, const float* dQ, int ldq, const float* dK, int ldk, const float* dV,
                    int ldv, float* dO, int ldo, int Tq, int Tk, int H, float scale, int causal,
                    int chunk, const float* dP, int ldp, const float* du, const float* dv);
int ss_op_dwconv_bn_silu(void* stream, const float* dx, int ldx, float* dy, int ldy, const float* dwt,
                         int K, const float* mean, const float* var, const float* gamma,
                         const float* beta, float eps, int T, int C, int chunk);

#ifdef __cplusplus
}
#endif
#endif /* STREAMSPEECH_HIP_H */
