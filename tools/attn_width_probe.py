"""How fast is a decode self-attention launch ALONE on the chip at the row counts the schedules use?  (round 6: is a lone
quarter-batch launch -- 64 rows x 6 heads = 384 workgroups of 3 waves -- parallelism-limited?  the trace digest shows exactly
one attention kernel in flight 32 % of the headline's wall time.)  f32 caches, append launches at depth 513, cold K/V (the
launches cycle over enough cache copies to exceed the 256 MB Infinity Cache), HIP events around 40 launches."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mt3_amd import _lib  # noqa: E402

lib = _lib.load()
H, cap, n_keys = 6, 1024, 513
s = torch.cuda.current_stream().cuda_stream
for B in (256, 128, 64, 32):
    copies = max(4, int(1.2e9 // (2 * B * H * cap * 64 * 4)) + 1)
    kc = [torch.randn(B, H, cap, 64, device="cuda") for _ in range(copies)]
    vc = [torch.randn(B, H, cap, 64, device="cuda") for _ in range(copies)]
    qkv = torch.randn(B, 3 * H * 64, device="cuda") * 0.3
    out = torch.empty(B, H * 64, device="cuda")
    step = torch.full((B,), n_keys - 1, device="cuda", dtype=torch.int32)

    def launch(i):
        _lib.check(lib.mt3_op_decode_attention(_lib.MT3_F32, qkv.data_ptr(), 3 * H * 64, kc[i % copies].data_ptr(),
                                               vc[i % copies].data_ptr(), cap, qkv.data_ptr() + H * 64 * 4,
                                               qkv.data_ptr() + 2 * H * 64 * 4, 3 * H * 64, step.data_ptr(), 0,
                                               out.data_ptr(), B, H, s))
    for i in range(copies):
        launch(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    n = 40
    for i in range(n):
        launch(i)
    e1.record()
    e1.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    mb = 2.0 * B * H * 64 * 4 * n_keys / 1e6
    print("B = %3d (%4d workgroups): %.1f us per launch back to back, %.0f MB of K/V -> %.2f TB/s" % (B, B * H, us, mb, mb / us), flush=True)
    del kc, vc
