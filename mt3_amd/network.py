"""Encoder-decoder engine wrapper (mirror of mt3/network.py's public surface).

`T5Config` keeps the reference's field names (network.py:25-41; values of
mt3/gin/model.gin:47-59 as defaults).  `Transformer` owns one `mt3_engine` of
libmt3hip.so: `encode()` = network.Transformer.encode (network.py:275-301) plus the
hoisted cross-attention K/V; `decode()` = the cached greedy loop that t5x runs
around network.Transformer.decode (network.py:303-361).  Device memory for inputs
and outputs is plain torch CUDA tensors; all compute is in the library.
"""
from __future__ import annotations

import ctypes as C
import dataclasses
import math
from typing import Dict, Optional, Sequence

import numpy as np

from . import _lib


@dataclasses.dataclass(frozen=True)
class T5Config:
    vocab_size: int = 1536
    dtype: str = "bfloat16"          # MFMA operand type: 'bfloat16' (product) or 'float32' (parity path)
    emb_dim: int = 512
    num_heads: int = 6
    num_encoder_layers: int = 8
    num_decoder_layers: int = 8
    head_dim: int = 64
    mlp_dim: int = 1024
    mlp_activations: Sequence[str] = ("gelu", "linear")
    dropout_rate: float = 0.1        # unused at inference
    logits_via_embedding: bool = False
    input_depth: int = 512           # spectrograms.input_depth
    kv_dtype: str = ""               # "" = K/V caches in `dtype`; "fp8_e4m3" (with bfloat16): e4m3 rows + per-row scales
    dense_dtype: str = ""            # "" = dense layers in `dtype`; "fp8_e4m3" (with bfloat16): the encoder's dense layers and
                                     # the cross-K/V projections as MXFP8 (e4m3 + one E8M0 scale per 32 K) on the scaled MFMA


MT3_SMALL = T5Config()                                           # model.gin / ismir2022/small.gin
MT3_BASE = T5Config(emb_dim=768, num_heads=12, num_encoder_layers=12, num_decoder_layers=12,
                    mlp_dim=2048)                                # ismir2022/base.gin:4-10


def param_shapes(cfg: T5Config) -> Dict[str, tuple]:
    """Flax parameter tree of network.Transformer flattened with '/' (SURVEY.md A.3)."""
    e, hd, f, v = cfg.emb_dim, cfg.num_heads * cfg.head_dim, cfg.mlp_dim, cfg.vocab_size
    s = {"encoder/continuous_inputs_projection/kernel": (cfg.input_depth, e),
         "encoder/encoder_norm/scale": (e,),
         "decoder/token_embedder/embedding": (v, e),
         "decoder/decoder_norm/scale": (e,),
         "decoder/logits_dense/kernel": (e, v)}

    def attn(p):
        for n in ("query", "key", "value"):
            s[f"{p}/{n}/kernel"] = (e, hd)
        s[f"{p}/out/kernel"] = (hd, e)

    def mlp(p):
        s[f"{p}/wi_0/kernel"] = (e, f)
        s[f"{p}/wi_1/kernel"] = (e, f)
        s[f"{p}/wo/kernel"] = (f, e)

    for i in range(cfg.num_encoder_layers):
        p = f"encoder/layers_{i}"
        s[f"{p}/pre_attention_layer_norm/scale"] = (e,)
        attn(f"{p}/attention")
        s[f"{p}/pre_mlp_layer_norm/scale"] = (e,)
        mlp(f"{p}/mlp")
    for i in range(cfg.num_decoder_layers):
        p = f"decoder/layers_{i}"
        s[f"{p}/pre_self_attention_layer_norm/scale"] = (e,)
        attn(f"{p}/self_attention")
        s[f"{p}/pre_cross_attention_layer_norm/scale"] = (e,)
        attn(f"{p}/encoder_decoder_attention")
        s[f"{p}/pre_mlp_layer_norm/scale"] = (e,)
        mlp(f"{p}/mlp")
    return s


def init_random_params(cfg: T5Config, seed: int = 0, norm_scale_jitter: float = 0.0) -> Dict[str, np.ndarray]:
    """Random-init weights with the reference's initialisers:
    embedding N(0,1) (network.py:221); attention kernels N(0, 1/fan_in), query / sqrt(head_dim)
    (layers.py:177-178,230-234); MLP / logits / input projection truncated-normal fan-in
    (layers.py:384-385, flax lecun_normal); norm scales 1 (optionally jittered for tests so that
    the scale-folding is exercised)."""
    import torch
    g = torch.Generator().manual_seed(seed)
    out = {}
    for name, shape in param_shapes(cfg).items():
        if name.endswith("/scale"):
            w = torch.ones(shape)
            if norm_scale_jitter:
                w = w + norm_scale_jitter * torch.randn(shape, generator=g)
        elif name.endswith("/embedding"):
            w = torch.randn(shape, generator=g)
        else:
            fan_in = shape[0]
            std = math.sqrt(1.0 / fan_in)
            if "/attention/" in name or "_attention/" in name:
                w = torch.randn(shape, generator=g) * std
                if name.endswith("query/kernel"):
                    w = w / math.sqrt(cfg.head_dim)
            else:
                w = torch.empty(shape)
                torch.nn.init.trunc_normal_(w, mean=0.0, std=1.0, a=-2.0, b=2.0, generator=g)
                w = w * (std / 0.87962566103423978)
        out[name] = w.numpy().astype(np.float32)
    return out


class Transformer:
    """One engine per (device, stream)."""

    def __init__(self, config: T5Config, input_length: int = 256, max_decode_length: int = 1024,
                 max_batch: int = 8, decode_chains: int = 1, options: int = 0):
        """options: bit set of _lib.OPT_* (mt3_engine_config.options): how the same function is evaluated --
        OPT_SINGLE_RESIDUAL_STREAM / OPT_ENCODER_SINGLE_RESIDUAL_STREAM (no bf16 copy / partial sums of the decoder's /
        encoder's residual rows), OPT_SEPARATE_PROJECTIONS
        (no folding of projections into neighbouring launches).  0 = the defaults."""
        self.config = config
        self.input_length, self.max_decode_length, self.max_batch = input_length, max_decode_length, max_batch
        if config.dtype not in ("bfloat16", "float32"):
            raise ValueError("T5Config.dtype must be 'bfloat16' or 'float32'")
        if config.kv_dtype not in ("", "fp8_e4m3") or (config.kv_dtype and config.dtype != "bfloat16"):
            raise ValueError("T5Config.kv_dtype must be '' or 'fp8_e4m3' (the latter with dtype 'bfloat16')")
        if config.dense_dtype not in ("", "fp8_e4m3") or (config.dense_dtype and config.dtype != "bfloat16"):
            raise ValueError("T5Config.dense_dtype must be '' or 'fp8_e4m3' (the latter with dtype 'bfloat16')")
        self._lib = _lib.load()
        ec = _lib.EngineConfig(config.vocab_size, config.emb_dim, config.num_heads, config.head_dim, config.mlp_dim,
                               config.num_encoder_layers, config.num_decoder_layers, config.input_depth,
                               input_length, max_decode_length, max_batch,
                               _lib.MT3_BF16 if config.dtype == "bfloat16" else _lib.MT3_F32, decode_chains,
                               _lib.MT3_FP8_E4M3 if config.kv_dtype == "fp8_e4m3" else 0,
                               _lib.MT3_FP8_E4M3 if config.dense_dtype == "fp8_e4m3" else 0, options)
        self._ec = ec
        self._h = None
        self._create()

    def _create(self):
        h = C.c_void_p()
        _lib.check(self._lib.mt3_engine_create(C.byref(self._ec), C.byref(h)))
        old, self._h = self._h, h
        if old:
            self._lib.mt3_engine_destroy(old)
        self._loaded = False

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.mt3_engine_destroy(h)

    def load_params(self, params: Dict[str, np.ndarray]):
        """`params`: flat dict in the reference's names/orientation (f32).  Names outside the network's
        parameter tree (optimizer state, other heads of a larger checkpoint) are ignored; kernels stored with
        split head axes (exactly [in, heads, head_dim] or [heads, head_dim, out]) are flattened to the 2-D DenseGeneral
        form, any other shape is an error; names outside the tree are listed in `self.ignored_params`; a second call
        re-restores (the engine is rebuilt and the encoded batch of the old one is gone), as the reference's
        restore_from_checkpoint allows."""
        if self._loaded:
            self._create()
        want = param_shapes(self.config)
        missing = [n for n in want if n not in params]
        if missing:
            raise _lib.Mt3Error(_lib.MT3_ERR_MISSING, "weights missing from the checkpoint: %s%s"
                                % (", ".join(missing[:4]), " ..." if len(missing) > 4 else ""))
        self.ignored_params = sorted(n for n in params if n not in want)     # reported, not silently dropped
        H, D = self.config.num_heads, self.config.head_dim
        for name, shape in want.items():
            arr = np.asarray(params[name])
            if arr.shape != shape:
                # DenseGeneral kernels with split head axes (layers.py:373-418): ONLY the two layouts flax produces
                # -- q/k/v [in, heads, head_dim], out [heads, head_dim, out] -- whose C-order flattening is the 2-D form
                if arr.ndim == 3 and len(shape) == 2 and (arr.shape == (shape[0], H, D) or arr.shape == (H, D, shape[1])):
                    arr = arr.reshape(shape)
                else:
                    raise _lib.Mt3Error(_lib.MT3_ERR_INVALID, "weight %s has shape %s, expected %s"
                                        % (name, tuple(arr.shape), shape))
            a = np.ascontiguousarray(arr, dtype=np.float32)
            shape = (C.c_int64 * a.ndim)(*a.shape)
            _lib.check(self._lib.mt3_engine_load_weight(self._h, name.encode(), a.ctypes.data, shape, a.ndim))
        _lib.check(self._lib.mt3_engine_finalize(self._h))
        self._loaded = True

    @property
    def device_bytes(self) -> int:
        return int(self._lib.mt3_engine_device_bytes(self._h))

    def encode(self, encoder_input_tokens, return_encoded: bool = False):
        """encoder_input_tokens: CUDA f32 tensor [B, T, input_depth]."""
        import torch
        x = encoder_input_tokens
        if x.dim() != 3 or x.shape[1] != self.input_length or x.shape[2] != self.config.input_depth:
            raise ValueError(f"expected [B, {self.input_length}, {self.config.input_depth}], got {tuple(x.shape)}")
        x = x.to(device="cuda", dtype=torch.float32).contiguous()
        enc = torch.empty((x.shape[0], x.shape[1], self.config.emb_dim), device="cuda", dtype=torch.float32) \
            if return_encoded else None
        _lib.check(self._lib.mt3_engine_encode(self._h, x.data_ptr(), x.shape[0], enc.data_ptr() if enc is not None
                                               else None, torch.cuda.current_stream().cuda_stream))
        self._batch = x.shape[0]
        return enc

    def decode(self, num_steps: Optional[int] = None, use_graph: bool = True, early_exit: bool = False,
               return_first_logits: bool = False, chains: int = 0, beam1: bool = False, single_stream: bool = False,
               wait: bool = True):
        """Decode for the batch of the last `encode`: greedy until EOS, or with `beam1` the selection
        rule of t5x beam_search(num_decodes=1, alpha=0.6) that the reference's predict_tokens runs.
        Returns int32 CUDA [B, L] ids (and the step-0 logits [B, V] if asked).  Batches of >= 128 rows are
        decoded as 2 or 4 row groups on streams with hardware queues of their own (include/mt3_hip.h) unless `single_stream`.
        early_exit: stop when every row has finished, and RETIRE finished rows meanwhile (no K/V is streamed for them, the
        live rows are compacted; ids up to each row's EOS are unchanged).  wait=False: MT3_DECODE_ASYNC -- the call returns
        once the engine's worker threads have the decode; `decode_wait()` joins it and returns the ids."""
        import torch
        B, L = self._batch, self.max_decode_length
        ids = torch.empty((B, L), device="cuda", dtype=torch.int32)
        logits = torch.empty((B, self.config.vocab_size), device="cuda", dtype=torch.float32) \
            if return_first_logits else None
        flags = (0 if use_graph else _lib.DECODE_NO_GRAPH) | (_lib.DECODE_EARLY_EXIT if early_exit else 0) | \
            ((chains & 0xF) << 8) | (_lib.DECODE_BEAM1 if beam1 else 0) | \
            (_lib.DECODE_SINGLE_STREAM if single_stream else 0) | (0 if wait else _lib.DECODE_ASYNC)
        ran = C.c_int32()
        _lib.check(self._lib.mt3_engine_decode(self._h, B, num_steps or L, flags, ids.data_ptr(),
                                               logits.data_ptr() if logits is not None else None, C.byref(ran),
                                               torch.cuda.current_stream().cuda_stream))
        if not wait:
            self._async = (ids, logits)              # kept alive until decode_wait
            return None
        self.steps_run = ran.value
        return (ids, logits) if return_first_logits else ids

    def transcribe(self, encoder_input_tokens, num_steps: Optional[int] = None, beam1: bool = False,
                   use_graph: bool = True, single_stream: bool = False, debug_poll_steps: int = 0,
                   debug_row_groups: int = 0, debug_skip_encoder_passes: bool = False):
        """mt3_engine_transcribe: encode + decode of ANY number of segments through the engine's `max_batch` decode slots
        with in-flight batching -- a slot whose segment has finished restarts on the next one (the reference's loop over
        `.batch(8)` calls of predict_batch_with_aux, NB:295-301, without its batch-synchronous wait for the longest row).
        encoder_input_tokens: CUDA f32 [N, T, input_depth].  Returns int32 CUDA [N, L] ids, row i = segment i, bit-identical
        to encode() + decode(early_exit=True) of that segment; `self.transcribe_stats` says what ran."""
        import torch
        x = encoder_input_tokens
        if x.dim() != 3 or x.shape[1] != self.input_length or x.shape[2] != self.config.input_depth:
            raise ValueError(f"expected [N, {self.input_length}, {self.config.input_depth}], got {tuple(x.shape)}")
        x = x.to(device="cuda", dtype=torch.float32).contiguous()
        N, L = x.shape[0], self.max_decode_length
        ids = torch.empty((N, L), device="cuda", dtype=torch.int32)
        flags = (0 if use_graph else _lib.DECODE_NO_GRAPH) | (_lib.DECODE_BEAM1 if beam1 else 0) | \
            (_lib.DECODE_SINGLE_STREAM if single_stream else 0)
        st = _lib.TranscribeStats()
        if debug_poll_steps or debug_row_groups or debug_skip_encoder_passes:      # mt3_debug_engine_transcribe (A/B measurements only)
            _lib.check(self._lib.mt3_debug_engine_transcribe(self._h, x.data_ptr(), N, num_steps or L, flags,
                                                             debug_poll_steps, debug_row_groups,
                                                             1 if debug_skip_encoder_passes else 0, ids.data_ptr(),
                                                             C.byref(st), torch.cuda.current_stream().cuda_stream))
        else:
            _lib.check(self._lib.mt3_engine_transcribe(self._h, x.data_ptr(), N, num_steps or L, flags, ids.data_ptr(),
                                                       C.byref(st), torch.cuda.current_stream().cuda_stream))
        self._batch = min(N, self.max_batch)
        self.transcribe_stats = {n: int(getattr(st, n)) for n, _ in st._fields_ if n != "reserved"}
        self.steps_run = st.steps_run
        return ids

    def decode_wait(self):
        """mt3_engine_decode_wait: join the decode a `decode(wait=False)` started; returns what that call would have."""
        ran = C.c_int32()
        _lib.check(self._lib.mt3_engine_decode_wait(self._h, C.byref(ran)))
        ids, logits = self._async
        self._async = None
        self.steps_run = ran.value
        return (ids, logits) if logits is not None else ids

    def debug_set_eos_schedule(self, lengths=None):
        """mt3_debug_engine_set_eos_schedule (include/mt3_hip_debug.h): impose output lengths -- row r's distribution at
        step lengths[r] - 1 becomes a point mass on EOS (SURVEY.md 8(d)'s synthetic EOS schedule).  None: off."""
        if lengths is None:
            _lib.check(self._lib.mt3_debug_engine_set_eos_schedule(self._h, None, 0))
            return
        a = np.ascontiguousarray(lengths, dtype=np.int32)
        _lib.check(self._lib.mt3_debug_engine_set_eos_schedule(self._h, a.ctypes.data, int(a.size)))

    def debug_decode(self, num_steps: Optional[int] = None, skip_self_attn: bool = False,
                     skip_cross_attn: bool = False, chains: int = 0, use_graph: bool = True):
        """mt3_debug_engine_decode (include/mt3_hip_debug.h): a decode with kernels left out of every step, for
        differential timing only -- the ids it returns are meaningless."""
        import torch
        B, L = self._batch, self.max_decode_length
        ids = torch.empty((B, L), device="cuda", dtype=torch.int32)
        skip = (_lib.DEBUG_SKIP_SELF_ATTN if skip_self_attn else 0) | (_lib.DEBUG_SKIP_CROSS_ATTN if skip_cross_attn else 0)
        flags = (0 if use_graph else _lib.DECODE_NO_GRAPH) | ((chains & 0xF) << 8)
        _lib.check(self._lib.mt3_debug_engine_decode(self._h, B, num_steps or L, flags, skip, ids.data_ptr(),
                                                     torch.cuda.current_stream().cuda_stream))
        return ids

    def debug_poison_caches(self, pattern: int = 0xFF, cross: bool = False):
        """mt3_debug_engine_poison_caches: fill the K/V caches with a byte pattern (0xFF = NaN in every cache format)."""
        import torch
        _lib.check(self._lib.mt3_debug_engine_poison_caches(self._h, pattern, 1 if cross else 0,
                                                            torch.cuda.current_stream().cuda_stream))

    def decode_forced(self, forced_ids, num_steps: Optional[int] = None, use_graph: bool = True,
                      return_logits: bool = True, chains: int = 0):
        """Teacher-forced cached decode (network.Transformer.decode on given decoder inputs, one token per cached
        step): step 0 is fed BOS, step t+1 is fed forced_ids[:, t].  forced_ids: int32 [B, <= L].
        Returns (argmax ids int32 CUDA [B, L], logits f32 CUDA [num_steps, B, V] or None)."""
        import torch
        B, L = self._batch, self.max_decode_length
        f = torch.zeros((B, L), device="cuda", dtype=torch.int32)
        src = torch.as_tensor(forced_ids).to(device="cuda", dtype=torch.int32)
        if src.dim() != 2 or src.shape[0] != B or src.shape[1] > L:
            raise ValueError(f"forced_ids must be [B={B}, <= {L}], got {tuple(src.shape)}")
        f[:, : src.shape[1]] = src
        n = num_steps or L
        ids = torch.empty((B, L), device="cuda", dtype=torch.int32)
        logits = torch.empty((n, B, self.config.vocab_size), device="cuda", dtype=torch.float32) \
            if return_logits else None
        flags = (0 if use_graph else _lib.DECODE_NO_GRAPH) | ((chains & 0xF) << 8)
        _lib.check(self._lib.mt3_engine_decode_forced(self._h, B, n, flags, f.data_ptr(),
                                                      logits.data_ptr() if logits is not None else None,
                                                      ids.data_ptr(), torch.cuda.current_stream().cuda_stream))
        torch.cuda.current_stream().synchronize()          # `f` must outlive the copy the call enqueued
        return ids, logits

    def status(self, what: int) -> int:
        """mt3_engine_status: _lib.STATUS_* (graph fallbacks, row groups / compactions of the last decode, ...)."""
        rc = int(self._lib.mt3_engine_status(self._h, what))
        if rc < 0:
            _lib.check(rc)
        return rc
