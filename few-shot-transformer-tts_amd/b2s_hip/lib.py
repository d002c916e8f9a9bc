"""ctypes binding of libb2s_hip.so (C ABI in include/b2s_hip.h).

The library is the product; there is NO CPU fallback: if it cannot be loaded, or a tensor is not on
a HIP device, the calls raise.  PyTorch is used only for device memory, streams and autograd plumbing.
"""
import ctypes as C
import os

# The engine runs its weight-gradient GEMMs on a second HIP stream.  HIP maps streams onto GPU_MAX_HW_QUEUES hardware queues
# (default 4); with RCCL's streams in the process the second stream shared the main stream's queue and the two serialised
# (+10 % step time).  Effective only if the HIP runtime has not initialised yet -- importing this package before the first
# torch.cuda call is enough; launchers can also export it.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
# (RCCL's channel count is the LAUNCHER's setting, not this package's: every channel is a workgroup that holds a CU while a collective
# runs -- the step loses 4-5 % with 8-32 CUs held, 10.6 % with 64 (profiles/r03_cu_loss.txt) -- so bench.py exports NCCL_MAX_NCHANNELS=16
# before it creates its process group, and INTEGRATION.md section 5 recommends the same for train.py; importing this package no longer
# touches NCCL_* -- it would throttle every collective of the process, the caller's own included.)

import torch  # noqa: E402

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("B2S_LIB_PATH") or os.path.join(os.path.dirname(_HERE), "libb2s_hip.so")      # (override: instrumented development builds)

_lib = None


class B2SError(RuntimeError):
    pass


class Config(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "num_mels", "vocab_size", "embed_size", "encoder_hidden", "decoder_hidden",
        "n_encoder_layer", "n_decoder_layer", "n_attention_head",
        "prenet_hidden", "postnet_hidden", "n_postnet_layer",
        "multi_speaker", "max_num_speaker", "speaker_embedding_size",
        "multi_lingual", "max_num_language", "language_embedding_size")] + [
        ("transformer_dropout_rate", C.c_float), ("decoder_dropout_rate", C.c_float),
        ("reg_weight", C.c_float), ("compute_dtype", C.c_int32),
        ("guided_attention_weight", C.c_float), ("guided_attention_sigma", C.c_float), ("freeze_encoder", C.c_int32)]


class GemmDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("dtype", "trans_a", "trans_b", "M", "N", "K", "lda", "ldb", "ldc", "c_fp32",
                                         "batch", "batch_inner")] + \
               [(n, C.c_int64) for n in ("a_bs_o", "a_bs_i", "b_bs_o", "b_bs_i", "c_bs_o", "c_bs_i")] + \
               [("alpha", C.c_float), ("relu", C.c_int32), ("accumulate", C.c_int32), ("drop_p", C.c_float),
                ("seed", C.c_uint64), ("conv_cin_a", C.c_int32), ("conv_T", C.c_int32), ("conv_dw_cin", C.c_int32),
                ("rows_per_batch", C.c_int32)]


P = C.c_void_p
_PROTOS = {
    "b2s_last_error": (C.c_char_p, []),
    "b2s_version": (C.c_int, []),
    "b2s_model_create": (C.c_int, [C.POINTER(Config), C.POINTER(P)]),
    "b2s_model_destroy": (None, [P]),
    "b2s_model_num_tensors": (C.c_int, [P]),
    "b2s_model_tensor_info": (C.c_int, [P, C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "b2s_model_bind": (C.c_int, [P, C.POINTER(P), C.POINTER(P), C.c_int]),
    "b2s_model_sync_weights": (C.c_int, [P, P, C.c_int]),
    "b2s_encoder_ws_bytes": (C.c_size_t, [P, C.c_int, C.c_int]),
    "b2s_encoder_forward": (C.c_int, [P, P, P, P, P, C.c_int, C.c_int, C.c_int, C.c_uint64, P, C.c_size_t, P, P, C.POINTER(P)]),
    "b2s_encoder_backward": (C.c_int, [P, P, P, P]),
    "b2s_decoder_ws_bytes": (C.c_size_t, [P, C.c_int, C.c_int, C.c_int]),
    "b2s_decoder_forward": (C.c_int, [P, P, P, P, P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64, P, C.c_size_t, P, P, P, P, C.POINTER(P)]),
    "b2s_decoder_backward": (C.c_int, [P, P, P, P, P, C.c_int, P, P, P]),
    "b2s_decoder_guided_loss": (C.c_int, [P, P, P, P, P]),
    "b2s_decoder_alignment": (C.c_int, [P, P, C.c_int, C.c_int, P, P]),
    "b2s_postnet_ws_bytes": (C.c_size_t, [P, C.c_int, C.c_int]),
    "b2s_postnet_forward": (C.c_int, [P, P, P, P, C.c_int, C.c_int, C.c_int, C.c_uint64, P, C.c_size_t, P, P, C.POINTER(P)]),
    "b2s_postnet_backward": (C.c_int, [P, P, P, P, C.c_int, P]),
    "b2s_ctx_free": (None, [P]),
    "b2s_loss_forward": (C.c_int, [P, P, P, P, P, P, C.c_int, C.c_int, P, P, P, P]),
    "b2s_loss_backward": (C.c_int, [P, P, P, P, P, P, C.c_int, C.c_int, P, P, P, P, P]),
    "b2s_l2_backward": (C.c_int, [P, P, P]),
    "b2s_adam_bind": (C.c_int, [P, C.POINTER(P), C.POINTER(P), C.c_int]),
    "b2s_adam_step": (C.c_int, [P, C.c_float, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, P]),
    "b2s_zero_grads": (C.c_int, [P, P, C.c_int]),
    "b2s_model_set_grad_slot_padding": (C.c_int, [P, C.c_int]),
    "b2s_gemm": (C.c_int, [C.POINTER(GemmDesc), P, P, P, P, P, P, P, P]),
    "b2s_gemm_splitk": (C.c_int, [C.POINTER(GemmDesc), C.c_int, P, P, P, P, C.c_size_t, P]),
    "b2s_layernorm_forward": (C.c_int, [C.c_int, P, P, P, P, P, P, C.c_int, C.c_int, C.c_float, P]),
    "b2s_layernorm_backward": (C.c_int, [C.c_int, P, P, P, P, P, P, P, P, C.c_int, C.c_int, P]),
    "b2s_attention_ws_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "b2s_attention_forward": (C.c_int, [C.c_int, P, C.c_int, P, C.c_int, P, C.c_int, P, C.c_int, C.c_int, C.c_int, C.c_int,
                                        C.c_int, C.c_int, C.c_int, P, P, C.c_int64, C.c_int64, C.c_float, C.c_uint64, P, P, P, P]),
    "b2s_attention_backward": (C.c_int, [C.c_int, P, C.c_int, P, C.c_int, P, C.c_int, P, C.c_int, P, P, P, C.c_int, P, C.c_int,
                                         P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_uint64, P, P]),
    "b2s_flash_attention_forward": (C.c_int, [C.c_int, P, C.c_int, P, C.c_int, P, C.c_int, P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                              C.c_int, C.c_int, P, C.c_float, C.c_uint64, P, P]),
    "b2s_flash_attention_backward": (C.c_int, [C.c_int, P, P, C.c_int, P, C.c_int, P, C.c_int, P, C.c_int, P, P, P, C.c_int, P, C.c_int,
                                               P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, P, C.c_float,
                                               C.c_uint64, P]),
    "b2s_flash_attention_align": (C.c_int, [C.c_int, P, C.c_int, P, C.c_int, P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, P, P,
                                            P]),
    "b2s_align_from_probs": (C.c_int, [C.c_int, P, P, C.c_int, C.c_int, C.c_int, C.c_int, P]),
    "b2s_add3": (C.c_int, [P, P, P, P, C.c_int64, P]),
    "b2s_cast": (C.c_int, [C.c_int, P, P, C.c_int64, P]),
    "b2s_cast_back": (C.c_int, [C.c_int, P, P, C.c_int64, P]),
    "b2s_decode_ws_bytes": (C.c_size_t, [P, C.c_int, C.c_int, C.c_int, C.c_int]),
    "b2s_decode_begin": (C.c_int, [P, P, P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_int, P, C.c_size_t, P, C.POINTER(P)]),
    "b2s_decode_run": (C.c_int, [P, P, C.c_int, C.c_int, P]),
    "b2s_decode_status": (C.c_int, [P, C.POINTER(C.c_int), C.POINTER(C.c_int), P]),
    "b2s_decode_fetch": (C.c_int, [P, P, C.c_int, P, P, P]),
    "b2s_decode_alignment": (C.c_int, [P, P, C.c_int, C.c_int, C.c_int, P, P]),
    "b2s_decode_end": (None, [P]),
    "b2s_model_second_stream": (C.c_void_p, [P]),
    "b2s_model_set_side_stream": (C.c_int, [P, P]),
    "b2s_gemm_set_tile_policy": (C.c_int, [C.c_int]),
    "b2s_model_backward_abort": (C.c_int, [P, P]),
    "b2s_model_mark_grads_ready": (C.c_int, [P]),
    "b2s_adam_shard": (C.c_int, [P, P, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_int]),
    "b2s_param_wire": (C.c_int, [P, P, C.c_int, P]),
    "b2s_model_set_stage_hook": (C.c_int, [P, P, P, P]),
    "b2s_prof_enable": (None, [C.c_int]),
    "b2s_adam_set_grad_wire": (C.c_int, [P, P, P]),
    "b2s_adam_step_groups": (C.c_int, [P, C.c_float, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int, C.c_int, P]),
    "b2s_prof_collect": (C.c_int, [C.POINTER(C.c_double), C.c_int]),
    "b2s_dropout_mask": (C.c_int, [C.c_float, C.c_uint64, C.c_uint32, P, C.c_int64, P]),
    "b2s_decoder_compact_rows": (C.c_int, [P, P, C.c_int]),
    "b2s_dropout_mask_attn": (C.c_int, [C.c_float, C.c_uint64, C.c_uint32, P, C.c_int64, C.c_int, P]),
    "b2s_dropout_site": (C.c_int, [C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_uint32), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "b2s_encf_attention_forward": (C.c_int, [P, P, P, P, C.c_int, C.c_int, C.c_float, C.c_uint64, C.c_uint32, P, P, P, P, C.c_int, P]),
    "b2s_encf_attention_backward": (C.c_int, [P, P, P, P, P, P, P, C.c_int, C.c_int, C.c_float, C.c_uint64, C.c_uint32, P, P, C.c_int, P]),
    "b2s_encf_ffn_sublayer": (C.c_int, [C.c_int, P, P, P, P, P, C.c_int, C.c_int, C.c_float, C.c_uint64, C.c_uint32, P, C.c_int, P]),
    "b2s_encf_reduce_layernorm_forward": (C.c_int, [P, P, C.c_int, C.c_int, C.c_float, C.c_uint64, C.c_uint32, P, P, P, P, P, C.c_int, P, P, C.c_int, P]),
    "b2s_encf_reduce_layernorm_backward": (C.c_int, [P, C.c_int, C.c_int, P, P, P, P, P, P, P, P, P, C.c_float, C.c_uint64, C.c_uint32, C.c_int, P]),
    "b2s_transpose_bf16": (C.c_int, [P, P, C.c_int, C.c_int, P]),
}
EXPORTS = sorted(_PROTOS)


def load():
    """Load libb2s_hip.so (raises B2SError if it is missing -- there is no fallback path)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise B2SError("libb2s_hip.so not found at %s -- build it with few-shot-transformer-tts_amd/csrc/build.sh "
                       "(or __graft_entry__.build()); there is no CPU fallback" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _PROTOS.items():
        fn = getattr(lib, name)           # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        raise B2SError(load().b2s_last_error().decode("utf-8", "replace"))


def ptr(t):
    """Device pointer of a tensor (None -> NULL).  Refuses CPU tensors: the HIP path is the only path."""
    if t is None:
        return None
    if not t.is_cuda:
        raise B2SError("tensor is on %s; the byte2speech hot path runs on a HIP device only" % t.device)
    if not t.is_contiguous():
        raise B2SError("tensor must be contiguous")
    return t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream
