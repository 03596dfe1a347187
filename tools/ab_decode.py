"""A/B of decode-loop variants on one box, one process: for each named variant (dtype, engine options, debug knobs) build
an engine, encode the same 256 segments, time the 1024-step graph-replayed greedy decode with HIP events (min of
REPS), and print ms / audio-s/s-equivalent.  Knobs are process-wide and baked into captured graphs, so every variant
gets a fresh engine.  Usage: python tools/ab_decode.py [variant-name-substring ...]"""
import dataclasses
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mt3_amd import _lib, network, spectrograms, synthetic  # noqa: E402

B = int(os.environ.get("AB_B", "256"))
REPS = int(os.environ.get("AB_REPS", "2"))
lib = _lib.load()
K = _lib
VARIANTS = [
    # name, dtype, kv, model, options, {knob: value}
    ("bf16 default", "bfloat16", "", "mt3", 0, {}),
    ("bf16 fold launch on 32x64 tiles", "bfloat16", "", "mt3", 0, {K.DEBUG_KNOB_FOLD_WIDE_TILE: 1}),
    ("bf16 GEGLU on 32x32 two-wave tiles", "bfloat16", "", "mt3", 0, {K.DEBUG_KNOB_GEGLU_NARROW_TILE: 1}),
    ("bf16 default (again)", "bfloat16", "", "mt3", 0, {}),
    ("bf16 two slices in flight", "bfloat16", "", "mt3", 0, {K.DEBUG_KNOB_PREFETCH2: 1}),
    ("bf16 xcd never n-major", "bfloat16", "", "mt3", 0, {K.DEBUG_KNOB_XCD_N_MAJOR: 2}),
    ("bf16 q-fold only (r2)", "bfloat16", "", "mt3", K.OPT_SEPARATE_QKV_PROJECTION, {}),
    ("bf16 q-fold only, two slices in flight", "bfloat16", "", "mt3", K.OPT_SEPARATE_QKV_PROJECTION, {K.DEBUG_KNOB_PREFETCH2: 1}),
    ("bf16 separate projections", "bfloat16", "", "mt3", K.OPT_SEPARATE_PROJECTIONS, {}),
    ("f32 default (split form + folds)", "float32", "", "mt3", 0, {}),
    ("f32 fold launch on 32x64 tiles", "float32", "", "mt3", 0, {K.DEBUG_KNOB_FOLD_WIDE_TILE: 1}),
    ("f32 two slices in flight", "float32", "", "mt3", 0, {K.DEBUG_KNOB_PREFETCH2: 1}),
    ("f32 eight-wave split-K tiles", "float32", "", "mt3", 0, {K.DEBUG_KNOB_F32_SPLIT_K: 1}),
    ("f32 q-fold only", "float32", "", "mt3", K.OPT_SEPARATE_QKV_PROJECTION, {}),
    ("f32 xcd never n-major", "float32", "", "mt3", 0, {K.DEBUG_KNOB_XCD_N_MAJOR: 2}),
    ("f32 separate projections", "float32", "", "mt3", K.OPT_SEPARATE_PROJECTIONS, {}),
    ("f32 r2 path", "float32", "", "mt3", K.OPT_SEPARATE_PROJECTIONS | K.OPT_SINGLE_RESIDUAL_STREAM,
     {}),
    ("fp8kv default", "bfloat16", "fp8_e4m3", "mt3", 0, {}),
    ("fp8kv q-fold only (r2)", "bfloat16", "fp8_e4m3", "mt3", K.OPT_SEPARATE_QKV_PROJECTION, {}),
    ("base fp8kv default", "bfloat16", "fp8_e4m3", "base", 0, {}),
    ("base fp8kv xcd never n-major", "bfloat16", "fp8_e4m3", "base", 0, {K.DEBUG_KNOB_XCD_N_MAJOR: 2}),
    ("base fp8kv K=768 always one slice", "bfloat16", "fp8_e4m3", "base", 0, {K.DEBUG_KNOB_NO_K768_SPLIT: 1}),
]
want = sys.argv[1:]
stream = torch.cuda.Stream()
audio = synthetic.synth_audio(B, seed=1000)
lm = spectrograms.compute_spectrogram_batch(audio, None)
ALL_KNOBS = (K.DEBUG_KNOB_DEC_ATTN_WAVES, K.DEBUG_KNOB_DEC_ATTN_FP8_WAVES, K.DEBUG_KNOB_NO_LDS_DMA_GEMM,
             K.DEBUG_KNOB_F32_SPLIT_K, K.DEBUG_KNOB_XCD_N_MAJOR, K.DEBUG_KNOB_PREFETCH2, K.DEBUG_KNOB_NO_K768_SPLIT,
             K.DEBUG_KNOB_FOLD_WIDE_TILE, K.DEBUG_KNOB_GEGLU_NARROW_TILE)
for name, dtype, kv, model, opt, knobs in VARIANTS:
    if want and not any(w in name for w in want):
        continue
    for k in ALL_KNOBS:
        _lib.check(lib.mt3_debug_set_knob(k, knobs.get(k, 0)))
    shape = network.MT3_BASE if model == "base" else network.MT3_SMALL
    cfg = dataclasses.replace(shape, dtype=dtype, kv_dtype=kv)
    eng = network.Transformer(cfg, input_length=256, max_decode_length=1024, max_batch=B, options=opt)
    eng.load_params(network.init_random_params(cfg, seed=0))
    best, noattn, prod = 1e30, 1e30, 1e30
    with torch.cuda.stream(stream):
        eng.encode(lm)
        eng.decode(num_steps=2, single_stream=True)
        eng.debug_decode(num_steps=2, skip_self_attn=True, skip_cross_attn=True)
        for _ in range(REPS):
            e0, e1, e2, e3 = (torch.cuda.Event(enable_timing=True) for _ in range(4))
            e0.record(stream)
            eng.decode(num_steps=1024, single_stream=True)
            e1.record(stream)
            eng.debug_decode(num_steps=1024, skip_self_attn=True, skip_cross_attn=True)
            e2.record(stream)
            eng.decode(num_steps=1024)                      # the schedule mt3_engine_decode picks (partitioned for bf16)
            e3.record(stream)
            e3.synchronize()
            best, noattn = min(best, e0.elapsed_time(e1)), min(noattn, e1.elapsed_time(e2))
            prod = min(prod, e2.elapsed_time(e3))
    print("%-52s one stream %8.1f ms | without attention %7.1f ms = %6.1f us/step | as shipped (%d groups) %8.1f ms = %7.1f "
          "audio-s/s decode-only" % (name, best, noattn, noattn * 1e3 / 1024, eng.status(K.STATUS_LAST_DECODE_GROUPS), prod,
                                     B * 2.048 / (prod * 1e-3)), flush=True)
    del eng
for k in ALL_KNOBS:
    lib.mt3_debug_set_knob(k, 0)
