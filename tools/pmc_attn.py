"""PMC pass for the roofline's `traffic` fields: run the decode attention kernels standalone at the bench's shape
(B=256, H=6) for a few key counts -- bf16, f32 and e4m3 caches, self (append) and cross (256 keys) -- plus a
calibration copy of known size and the log-mel frontend, so that `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE`
(separate passes) can be read per dispatch.
Usage (on the GPU box):  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -o f -- python tools/pmc_attn.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mt3_amd import _lib  # noqa: E402

lib = _lib.load()
B, H, cap = 256, 6, 1024
dev = "cuda"
s = torch.cuda.current_stream().cuda_stream
layers = 4                                     # 4 x (403 / 806 / 202 MB) of K+V: every launch streams cold data (> 256 MB MALL)
# calibration: a 1 GiB f32 copy = 1 GiB read + 1 GiB written, wide coalesced
a = torch.empty(256 * 1024 * 1024, device=dev)
a.normal_()
for _ in range(2):
    b = a.clone()
torch.cuda.synchronize()
del a, b
KEYS = (1024, 513, 129, 1024, 513, 129)


def self_and_cross(kind, H=H):
    tdt = torch.float32 if kind == "f32" else torch.bfloat16
    es = 4 if kind == "f32" else 2
    dt = _lib.MT3_F32 if kind == "f32" else _lib.MT3_BF16
    qkv = (torch.randn(B, 3 * H * 64, device=dev) * 0.3).to(tdt)
    q = (torch.randn(B, H * 64, device=dev) * 0.3).to(tdt)
    out = torch.empty(B, H * 64, device=dev, dtype=tdt)
    if kind == "fp8":
        kc = [torch.randint(0, 120, (B, H, cap, 64), device=dev, dtype=torch.uint8) for _ in range(layers)]
        vc = [torch.randint(0, 120, (B, H, cap, 64), device=dev, dtype=torch.uint8) for _ in range(layers)]
        sc = [torch.full((B, H, cap, 2), 2.0 ** -7, device=dev) for _ in range(layers)]
        ck = torch.randint(0, 120, (8, 2, B, H, 256, 64), device=dev, dtype=torch.uint8)
        cs = torch.full((8, B, H, 256, 2), 2.0 ** -7, device=dev)
    else:
        kc = [torch.randn(B, H, cap, 64, device=dev).to(tdt) for _ in range(layers)]
        vc = [torch.randn(B, H, cap, 64, device=dev).to(tdt) for _ in range(layers)]
        ck = torch.randn(8, 2, B, H, 256, 64, device=dev).to(tdt)
    torch.cuda.synchronize()
    for n_keys in KEYS:
        step = torch.full((B,), n_keys - 1, device=dev, dtype=torch.int32)      # per-row position counters
        for l in range(layers):
            if kind == "fp8":
                _lib.check(lib.mt3_op_decode_attention_fp8(qkv.data_ptr(), 3 * H * 64, kc[l].data_ptr(), vc[l].data_ptr(),
                                                           sc[l].data_ptr(), cap, qkv.data_ptr() + H * 64 * es,
                                                           qkv.data_ptr() + 2 * H * 64 * es, 3 * H * 64, step.data_ptr(),
                                                           0, out.data_ptr(), B, H, s))
            else:
                _lib.check(lib.mt3_op_decode_attention(dt, qkv.data_ptr(), 3 * H * 64, kc[l].data_ptr(),
                                                       vc[l].data_ptr(), cap, qkv.data_ptr() + H * 64 * es,
                                                       qkv.data_ptr() + 2 * H * 64 * es, 3 * H * 64, step.data_ptr(), 0,
                                                       out.data_ptr(), B, H, s))
        torch.cuda.synchronize()
    # cross-attention shape (256 keys, no append)
    for l in range(8):
        if kind == "fp8":
            _lib.check(lib.mt3_op_decode_attention_fp8(q.data_ptr(), H * 64, ck[l, 0].data_ptr(), ck[l, 1].data_ptr(),
                                                       cs[l].data_ptr(), 256, None, None, 0, None, 256, out.data_ptr(),
                                                       B, H, s))
        else:
            _lib.check(lib.mt3_op_decode_attention(dt, q.data_ptr(), H * 64, ck[l, 0].data_ptr(), ck[l, 1].data_ptr(),
                                                   256, None, None, 0, None, 256, out.data_ptr(), B, H, s))
    torch.cuda.synchronize()


for kind in ("bf16", "f32", "fp8"):
    self_and_cross(kind)
    torch.cuda.empty_cache()
# BASELINE configs[4] (ismir2022/base.gin shape): 12 heads, e4m3 caches (the summary tells the two fp8 passes apart by grid size)
self_and_cross("fp8", H=12)
torch.cuda.empty_cache()
# log-mel frontend at the bench shape (256 full segments), 3 launches on fresh inputs (north_star: rocprof counters
# report the frontend's HBM traffic; algorithmic = 655,360 B per segment)
from mt3_amd import spectrograms, synthetic  # noqa: E402
for i in range(3):
    au = synthetic.synth_audio(256, seed=50 + i)
    torch.cuda.synchronize()
    lm = spectrograms.compute_spectrogram_batch(au, None)
    torch.cuda.synchronize()
print("pmc_attn done")
