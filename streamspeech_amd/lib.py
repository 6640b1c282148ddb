"""ctypes binding of libstreamspeech_hip.so (include/streamspeech_hip.h).

The product path has NO CPU fallback: if the shared library is missing or a symbol declared in
the header is not exported, import of the engine fails loudly.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SS_HIP_LIB") or os.path.join(_HERE, "libstreamspeech_hip.so")   # env override: tuning builds (tools/)


class SSConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "input_feat", "conv_channels", "conv_kernel", "enc_dim", "enc_ffn", "enc_heads", "enc_layers",
        "dw_kernel", "src_vocab", "tgt_vocab", "mt_layers", "dec_dim", "dec_ffn", "dec_heads",
        "t2u_layers", "unit_layers", "unit_vocab", "ctc_upsample", "pad", "eos", "unk",
        "max_rel_pos", "max_tgt_pos")]


class SSVocoderConfig(C.Structure):
    _fields_ = [
        ("num_embeddings", C.c_int32), ("embedding_dim", C.c_int32), ("model_in_dim", C.c_int32),
        ("upsample_initial_channel", C.c_int32),
        ("n_up", C.c_int32), ("upsample_rates", C.c_int32 * 8), ("upsample_kernel_sizes", C.c_int32 * 8),
        ("n_res", C.c_int32), ("resblock_kernel_sizes", C.c_int32 * 4), ("resblock_dilations", (C.c_int32 * 3) * 4),
        ("dur_hidden", C.c_int32), ("dur_kernel", C.c_int32)]


_vp, _i, _f, _i64 = C.c_void_p, C.c_int, C.c_float, C.c_int64

# symbol -> (restype, argtypes); must list every function include/streamspeech_hip.h declares
SIGNATURES = {
    "ss_abi_version": (_i, []),
    "ss_error_string": (C.c_char_p, [_i]),
    "ss_model_create": (_i, [C.POINTER(SSConfig), _vp, C.c_size_t, C.POINTER(C.c_char_p), C.POINTER(_i64),
                             C.POINTER(_i64), _i, C.POINTER(_vp)]),
    "ss_model_destroy": (None, [_vp]),
    "ss_scratch_create": (_i, [C.POINTER(_vp)]),
    "ss_scratch_destroy": (None, [_vp]),
    "ss_scratch_set_cap": (_i, [_vp, C.c_size_t]),
    "ss_scratch_trim": (_i, [_vp, C.c_size_t]),
    "ss_scratch_bytes": (C.c_size_t, [_vp]),
    "ss_model_bind_scratch": (_i, [_vp, _vp]),
    "ss_vocoder_bind_scratch": (_i, [_vp, _vp]),
    "ss_fbank_num_frames": (_i, [_i]),
    "ss_fbank_cmvn": (_i, [_vp, _vp, _vp, _i, _f, _vp, C.POINTER(_i)]),
    "ss_encoder_out_len": (_i, [_i]),
    "ss_resample": (_i, [_vp, _vp, _i64, _i, _i, _vp, _i, _vp, _i64]),
    "ss_row_max_logprob": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "ss_log_softmax": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "ss_encoder_forward": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp]),
    "ss_encoder_stream_reset": (_i, [_vp]),
    "ss_encoder_stream_set_tail": (_i, [_vp, _i]),
    "ss_encoder_stream_set_deferred": (_i, [_vp, _i]),
    "ss_encoder_stream_status": (_i, [_vp, _vp, _vp]),
    "ss_debug_enc_step_inject_timeout": (_i, [_vp]),
    "ss_encoder_stream_forward": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "ss_ctc_greedy": (_i, [_vp, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp]),
    "ss_mt_begin": (_i, [_vp, _vp, _vp, _i]),
    "ss_mt_set_persistent": (_i, [_vp, _i]),
    "ss_model_set_pack_invariant": (_i, [_vp, _i]),
    "ss_model_get_pack_invariant": (_i, [_vp]),
    "ss_debug_canon": (_i, [_i]),
    "ss_mt_get_persistent": (_i, [_vp]),
    "ss_debug_mt_inject_timeout": (_i, [_vp]),
    "ss_mt_append": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _i]),
    "ss_mt_truncate": (_i, [_vp, _i]),
    "ss_mt_greedy": (_i, [_vp, _vp, _vp, _i, C.POINTER(C.c_int32), _i, _i, _i, C.POINTER(C.c_int32), C.POINTER(_i),
                          _vp, C.POINTER(_i)]),
    "ss_t2u_units": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _i]),
    "ss_vocoder_create": (_i, [C.POINTER(SSVocoderConfig), _vp, C.c_size_t, C.POINTER(C.c_char_p),
                               C.POINTER(_i64), C.POINTER(_i64), _i, C.POINTER(_vp)]),
    "ss_vocoder_destroy": (None, [_vp]),
    "ss_vocoder_set_bf16x3": (_i, [_vp, _i]),
    "ss_vocoder_forward": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp, _i64, _vp, C.POINTER(_i64)]),
    "ss_batch_fbank_cmvn": (_i, [_vp, _vp, _i, _vp, C.POINTER(_i64), C.POINTER(C.c_int32), _f, _vp, C.POINTER(C.c_int32)]),
    "ss_batch_encoder_forward": (_i, [_vp, _vp, _i, _vp, C.POINTER(C.c_int32), _i, _i, _vp, C.POINTER(C.c_int32)]),
    "ss_batch_ctc_greedy": (_i, [_vp, _vp, _i, _i, _vp, C.POINTER(C.c_int32), _vp, _vp, _vp, _vp]),
    "ss_batch_mt_greedy": (_i, [_vp, _vp, _i, _vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32), _i,
                                C.POINTER(C.c_int32), _i, C.POINTER(C.c_int32), _vp, _i]),
    "ss_batch_t2u_units": (_i, [_vp, _vp, _i, _vp, _i, C.POINTER(C.c_int32), _i, _i, _vp, _vp, _vp]),
    "ss_batch_vocoder_forward": (_i, [_vp, _vp, _i, _vp, C.POINTER(C.c_int32), _i, _vp, _vp, _i64, _vp,
                                      C.POINTER(_i64), C.POINTER(_i64)]),
    "ss_prof_enable": (_i, [_i]),
    "ss_prof_reset": (_i, []),
    "ss_prof_read": (_i, [_i, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(_i64), C.POINTER(C.c_double)]),
    "ss_prof_totals": (_i, [_i, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(_i64)]),
    "ss_prof_read_issued": (_i, [_i, C.POINTER(C.c_double)]),
    "ss_prof_shape_log": (_i, [_i]),
    "ss_prof_shape_dump": (_i, [C.c_char_p, _i]),
    "ss_prof_num_classes": (_i, []),
    "ss_prof_class_name": (C.c_char_p, [_i]),
    "ss_debug_force_tile": (_i, [_i, _i, _i]),
    "ss_op_ffn_fused": (_i, [_vp, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _i, _i, _i]),
    "ss_debug_ffn": (_i, [_i, _i, _i]),
    "ss_debug_rtlin": (_i, [_i, _i]),
    "ss_debug_conv_c64": (_i, [_i]),
    "ss_debug_conv_c32": (_i, [_i]),
    "ss_debug_conv_c16": (_i, [_i]),
    "ss_debug_enc_step_launches": (_i64, []),
    "ss_op_ln_linear": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _f, _i]),
    "ss_debug_last_logits": (_i, [_vp, _vp, _vp, _i64, C.POINTER(_i), C.POINTER(_i)]),
    "ss_debug_sk_errors": (_i, []),
    "ss_debug_attention_split": (_i, [_i]),
    "ss_debug_attention_q16": (_i, [_i]),
    "ss_op_conv_gemm": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i,
                             _i, _f, _i, _f, _f, _i]),
    "ss_op_layernorm": (_i, [_vp, _vp, _i, _vp, _i, _vp, _vp, _i, _i, _f]),
    "ss_op_attention": (_i, [_vp, _vp, _i, _vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _f, _i, _i, _vp, _i, _vp, _vp]),
    "ss_op_dwconv_bn_silu": (_i, [_vp, _vp, _i, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _f, _i, _i, _i]),
}

_lib = None


class StreamSpeechHipError(RuntimeError):
    pass


def load():
    """dlopen the HIP library and bind every declared symbol; raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise StreamSpeechHipError(
            f"{LIB_PATH} not found: build it with streamspeech_amd/csrc/build.sh "
            "(or __graft_entry__.build()).  There is no CPU fallback for the product path.")
    # torch FIRST: its wheel carries its own HIP runtime; libstreamspeech_hip.so must bind to the runtime torch initialised (device
    # memory, streams and events cross the boundary), not bring up /opt/rocm's copy before torch loads -- a process that dlopen-ed this
    # library and only then imported torch (`__graft_entry__.build()` followed by `smoke()` in one process) got "no ROCm-capable device"
    # from hipMalloc inside ss_model_create (profiles/r05_smoke_load_order.log).
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise StreamSpeechHipError(f"{LIB_PATH} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    if lib.ss_abi_version() != 2:
        raise StreamSpeechHipError("ABI version mismatch")
    _lib = lib
    return lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().ss_error_string(rc).decode()
        raise StreamSpeechHipError(f"{what} failed: {msg} (code {rc})")
