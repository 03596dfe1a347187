"""Encoder A/B on one box: encode() of 256 (and 64) segments with the 256 x 128 LDS-DMA GEMM tile against the 128 x 128
one (debug knob), bf16 and MXFP8 dense layers; HIP-event time per encode and TFLOP/s (12.214 GFLOP per segment incl.
the cross-K/V projections).  Under `rocprofv3 --kernel-trace --stats` the per-kernel averages of both variants show up
side by side (the tile height is a template argument of mt3k::gemm_glds_kernel)."""
import dataclasses
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mt3_amd import _lib, network, spectrograms, synthetic  # noqa: E402

lib = _lib.load()
x256 = spectrograms.compute_spectrogram_batch(synthetic.synth_audio(256, seed=3), None)
for dense in ("", "fp8_e4m3"):
    for no256, attn4 in (((0, 0), (1, 0), (0, 1), (2, 0)) if not dense else ((0, 0),)):
        _lib.check(lib.mt3_debug_set_knob(_lib.DEBUG_KNOB_NO_GLDS_256, 1 if no256 == 1 else 0))
        _lib.check(lib.mt3_debug_set_knob(_lib.DEBUG_KNOB_GLDS_FRAG_DB, 1 if no256 == 2 else 0))
        _lib.check(lib.mt3_debug_set_knob(_lib.DEBUG_KNOB_ENC_ATTN_4_WAVES, attn4))
        cfg = dataclasses.replace(network.T5Config(dtype="bfloat16"), dense_dtype=dense)
        eng = network.Transformer(cfg, input_length=256, max_decode_length=1024, max_batch=256)
        eng.load_params(network.init_random_params(cfg, seed=0))
        for B in (256, 64):
            x = x256[:B].contiguous()
            eng.encode(x)
            best = 1e30
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    eng.encode(x)
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / 5)
            print("%-8s %-24s %-28s B=%3d: %.3f ms  %.0f TF/s" % (
                "mxfp8" if dense else "bf16", ("256x128 tiles (default)", "128x128 tiles", "128x128, fragment DB")[no256],
                "attention 4 waves" if attn4 else "attention 8 waves (default)", B, best, 12.214 * B / best), flush=True)
        del eng
lib.mt3_debug_set_knob(_lib.DEBUG_KNOB_NO_GLDS_256, 0)
lib.mt3_debug_set_knob(_lib.DEBUG_KNOB_ENC_ATTN_4_WAVES, 0)
lib.mt3_debug_set_knob(_lib.DEBUG_KNOB_GLDS_FRAG_DB, 0)
