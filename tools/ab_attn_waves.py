"""EXPERIMENT (round 4): waves per (row, head) workgroup of the decode attention.  (A) the kernel alone through the C ABI at
B = 64 / 128 / 256 rows (the quarter- / half- / full-batch launches of the row-group schedule), 513 keys, rotating over four
cache sets so that nothing is Infinity-Cache resident; (B) whole decodes in the product schedule (direct launches: a
captured graph would keep the geometry it was captured with)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mt3_amd import _lib, network, spectrograms, synthetic  # noqa: E402

lib = _lib.load()
H, cap, n_keys = 6, 1024, 513
s = torch.cuda.current_stream().cuda_stream
for kind in ("f32", "bf16"):
    tdt = torch.float32 if kind == "f32" else torch.bfloat16
    es = 4 if kind == "f32" else 2
    dt = _lib.MT3_F32 if kind == "f32" else _lib.MT3_BF16
    for B in (64, 128, 256):
        sets = 4 if B > 64 else 8
        kc = [torch.randn(B, H, cap, 64, device="cuda").to(tdt) for _ in range(sets)]
        vc = [torch.randn(B, H, cap, 64, device="cuda").to(tdt) for _ in range(sets)]
        qkv = (torch.randn(B, 3 * H * 64, device="cuda") * 0.3).to(tdt)
        out = torch.empty(B, H * 64, device="cuda", dtype=tdt)
        step = torch.full((B,), n_keys - 1, device="cuda", dtype=torch.int32)
        alg = B * H * (2 * (n_keys - 1) * 64 * es + 6 * 64 * es)
        line = "%s self-attention alone, B=%3d (%4d workgroups), %d keys:" % (kind, B, B * H, n_keys)
        for nw in (3, 4, 6, 8):
            _lib.check(lib.mt3_debug_set_attn_waves(nw))

            def run(n):
                for i in range(n):
                    l = i % sets
                    _lib.check(lib.mt3_op_decode_attention(dt, qkv.data_ptr(), 3 * H * 64, kc[l].data_ptr(), vc[l].data_ptr(), cap,
                                                           qkv.data_ptr() + H * 64 * es, qkv.data_ptr() + 2 * H * 64 * es,
                                                           3 * H * 64, step.data_ptr(), 0, out.data_ptr(), B, H, s))
            run(sets)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            run(10 * sets)
            e1.record()
            e1.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / (10 * sets)
            line += "  nw=%d %.2f us (%.2f TB/s)" % (nw, us, alg / us / 1e6)
        print(line, flush=True)
        del kc, vc
        torch.cuda.empty_cache()

audio = synthetic.synth_audio(256, seed=1000)
for dtype in ("float32", "bfloat16"):
    cfg = network.T5Config(dtype=dtype)
    eng = network.Transformer(cfg, input_length=256, max_decode_length=1024, max_batch=256)
    eng.load_params(network.init_random_params(cfg, seed=0))
    eng.encode(spectrograms.compute_spectrogram_batch(audio, None))
    line = "%s whole decode, B=256, product schedule, direct launches:" % dtype
    ref = None
    for nw in (3, 4, 6, 8):
        _lib.check(lib.mt3_debug_set_attn_waves(nw))
        eng.decode(num_steps=8, use_graph=False)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(2):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ids = eng.decode(num_steps=1024, use_graph=False)
            e1.record()
            e1.synchronize()
            best = min(best, e0.elapsed_time(e1))
        if ref is None:
            ref = ids.clone()
        same = float((ids == ref).all(1).float().mean())
        line += "  nw=%d %.1f ms (rows identical to nw=3: %.3f)" % (nw, best, same)
    print(line, flush=True)
    del eng
_lib.check(lib.mt3_debug_set_attn_waves(0))
