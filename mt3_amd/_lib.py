"""ctypes binding of libmt3hip.so (the C ABI in include/mt3_hip.h).

The library is the product; there is NO Python/CPU fallback: if it is missing or a
call fails, the caller gets an exception (`Mt3Error`).  `import torch` happens
first on purpose so that the library binds to the same HIP runtime
(libamdhip64.so.7) that owns torch's device allocations.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libmt3hip.so")

MT3_OK, MT3_ERR_INVALID, MT3_ERR_HIP, MT3_ERR_CAPACITY, MT3_ERR_MISSING = 0, -1, -2, -3, -4
MT3_BF16, MT3_F32, MT3_FP8_E4M3 = 0, 1, 2
EPI_STORE, EPI_RESID, EPI_GEGLU, EPI_POS, EPI_F32, EPI_HEADS = range(6)
DECODE_NO_GRAPH, DECODE_EARLY_EXIT, DECODE_BEAM1, DECODE_SINGLE_STREAM, DECODE_ASYNC = 1, 2, 4, 8, 16
(OPT_SINGLE_RESIDUAL_STREAM, OPT_SEPARATE_PROJECTIONS, OPT_ENCODER_SINGLE_RESIDUAL_STREAM,
 OPT_SEPARATE_QKV_PROJECTION, OPT_NO_ROW_GROUPS) = 1, 2, 4, 8, 16              # mt3_engine_config.options
OPT_ENCODER_F32_MFMA = 32
OPT_SPIN_WAITS = 64
# include/mt3_hip_debug.h (measurement / fault injection; not the product ABI)
DEBUG_SKIP_SELF_ATTN, DEBUG_SKIP_CROSS_ATTN = 1, 2
(STATUS_GRAPH_FALLBACKS, STATUS_LAST_DECODE_USED_GRAPH, STATUS_RESIDUAL_SPLIT, STATUS_KV_FP8, STATUS_Q_FOLD,
 STATUS_DENSE_FP8, STATUS_QKV_FOLD, STATUS_LAST_DECODE_GROUPS, STATUS_PARTITION_FALLBACKS,
 STATUS_LAST_DECODE_COMPACTIONS) = range(10)
EV_SHIFT, EV_PITCH, EV_VELOCITY, EV_TIE, EV_PROGRAM, EV_DRUM = range(6)
EVENT_TYPE_NAMES = ("shift", "pitch", "velocity", "tie", "program", "drum")
SPEC_ONSETS, SPEC_NOTES, SPEC_TIES = range(3)


class Mt3Error(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libmt3hip error %d: %s" % (code, msg))
        self.code = code


class FrontendConfig(C.Structure):
    _fields_ = [("sample_rate", C.c_int32), ("hop_width", C.c_int32), ("num_mel_bins", C.c_int32),
                ("fft_size", C.c_int32), ("lo_hz", C.c_float), ("hi_hz", C.c_float), ("table_dtype", C.c_int32)]


class EngineConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "vocab_size", "emb_dim", "num_heads", "head_dim", "mlp_dim", "num_encoder_layers",
        "num_decoder_layers", "input_depth", "input_length", "max_decode_len", "max_batch", "compute_dtype",
        "decode_chains", "kv_cache_dtype", "dense_dtype", "options")]


class TranscribeStats(C.Structure):            # mt3_transcribe_stats
    _fields_ = [(n, C.c_int32) for n in ("slots", "groups", "steps_run", "polls", "refills", "starved_polls",
                                         "encoder_chunks", "compactions", "used_graph")] + [("reserved", C.c_int32 * 7)]


class EventRange(C.Structure):
    _fields_ = [("type", C.c_int32), ("min_value", C.c_int32), ("max_value", C.c_int32)]


class CodecDesc(C.Structure):
    _fields_ = [("steps_per_second", C.c_double), ("num_ranges", C.c_int32), ("ranges", EventRange * 8)]


class NoteStruct(C.Structure):
    _fields_ = [("start_time", C.c_double), ("end_time", C.c_double), ("pitch", C.c_int32),
                ("velocity", C.c_int32), ("program", C.c_int32), ("is_drum", C.c_int32),
                ("instrument", C.c_int32), ("reserved", C.c_int32)]


# every symbol include/mt3_hip.h and include/mt3_hip_debug.h declare: (name, restype, argtypes)
_P = C.c_void_p
SIGNATURES = {
    "mt3_last_error": (C.c_char_p, []),
    "mt3_abi_version": (C.c_int, []),
    "mt3_frontend_create": (C.c_int, [C.POINTER(FrontendConfig), C.POINTER(_P)]),
    "mt3_frontend_destroy": (None, [_P]),
    "mt3_frontend_mel_matrix": (C.c_int, [_P, _P, C.POINTER(C.c_int64)]),
    "mt3_frontend_logmel": (C.c_int, [_P, _P, C.c_int32, C.c_int32, _P, _P, _P]),
    "mt3_frontend_logmel_dev": (C.c_int, [_P, _P, C.c_int32, C.c_int32, _P, _P, _P]),
    "mt3_engine_create": (C.c_int, [C.POINTER(EngineConfig), C.POINTER(_P)]),
    "mt3_engine_destroy": (None, [_P]),
    "mt3_engine_load_weight": (C.c_int, [_P, C.c_char_p, _P, C.POINTER(C.c_int64), C.c_int32]),
    "mt3_engine_finalize": (C.c_int, [_P]),
    "mt3_engine_device_bytes": (C.c_int64, [_P]),
    "mt3_engine_encode": (C.c_int, [_P, _P, C.c_int32, _P, _P]),
    "mt3_engine_decode": (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, _P, _P, C.POINTER(C.c_int32), _P]),
    "mt3_engine_decode_wait": (C.c_int, [_P, C.POINTER(C.c_int32)]),
    "mt3_engine_transcribe": (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int32, _P, C.POINTER(TranscribeStats), _P]),
    "mt3_engine_decode_forced": (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, _P]),
    "mt3_engine_status": (C.c_int, [_P, C.c_int32]),
    "mt3_debug_engine_decode": (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, _P]),
    "mt3_debug_engine_transcribe": (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P,
                                              C.POINTER(TranscribeStats), _P]),
    "mt3_debug_engine_poison_caches": (C.c_int, [_P, C.c_int32, C.c_int32, _P]),
    "mt3_debug_engine_set_eos_schedule": (C.c_int, [_P, _P, C.c_int32]),
    "mt3_ids_to_tokens": (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, _P, _P]),
    "mt3_op_gemm": (C.c_int, [C.c_int32, _P, C.c_int32, C.c_int32, _P, _P, C.c_int32, C.c_int32, C.c_int32,
                              C.c_int32, _P, C.c_int32, C.c_int32, _P]),
    "mt3_op_gemm_ex": (C.c_int, [C.c_int32, _P, C.c_int32, C.c_int32, _P, _P, C.c_int32, C.c_int32, C.c_int32,
                                 C.c_int32, _P, C.c_int32, C.c_int32, _P, _P, _P, _P]),
    "mt3_op_residual_split": (C.c_int, [C.c_int32, _P, _P, _P, C.c_int32, C.c_int32, _P]),
    "mt3_op_encoder_attention": (C.c_int, [C.c_int32, _P, _P, C.c_int32, C.c_int32, C.c_int32, _P]),
    "mt3_op_decode_attention": (C.c_int, [C.c_int32, _P, C.c_int32, _P, _P, C.c_int32, _P, _P, C.c_int32, _P,
                                          C.c_int32, _P, C.c_int32, C.c_int32, _P]),
    "mt3_op_decode_attention_fp8": (C.c_int, [_P, C.c_int32, _P, _P, _P, C.c_int32, _P, _P, C.c_int32, _P, C.c_int32,
                                              _P, C.c_int32, C.c_int32, _P]),
    "mt3_op_kv_quantize_fp8": (C.c_int, [_P, _P, _P, C.c_int32, _P]),
    "mt3_host_mx8_quantize": (C.c_int, [_P, C.c_int64, C.c_int64, _P, _P]),
    "mt3_op_mx8_quantize": (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, _P]),
    "mt3_op_gemm_mx8": (C.c_int, [_P, _P, _P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P,
                                  _P, _P]),
    "mt3_build_codec": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.POINTER(CodecDesc)]),
    "mt3_codec_num_classes": (C.c_int, [C.POINTER(CodecDesc)]),
    "mt3_codec_decode_event": (C.c_int, [C.POINTER(CodecDesc), C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "mt3_codec_encode_event": (C.c_int, [C.POINTER(CodecDesc), C.c_int32, C.c_int32, C.POINTER(C.c_int32)]),
    "mt3_notes_decode": (C.c_int, [C.POINTER(CodecDesc), C.c_int32, C.c_int32, _P, _P, _P, _P, _P, _P, C.c_int64,
                                   C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                                   C.POINTER(C.c_double)]),
}

_lib = None


def load():
    """Load (once) and type the library.  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise Mt3Error(MT3_ERR_MISSING,
                       "%s not found: build it with `python -m mt3_amd.build` "
                       "(hipcc --offload-arch=gfx950); there is no CPU fallback" % LIB_PATH)
    try:
        import torch  # noqa: F401  -- load torch's HIP runtime first so both share one libamdhip64
    except Exception:
        pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)       # AttributeError here == a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int):
    if rc != MT3_OK:
        raise Mt3Error(rc, (load().mt3_last_error() or b"").decode("utf-8", "replace"))
    return rc
