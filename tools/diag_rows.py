import os, sys, math
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mt3_amd import _lib
lib = _lib.load()
s = lambda: torch.cuda.current_stream().cuda_stream
g = torch.Generator(device="cuda").manual_seed(0)
M, N, K = 64, 1536, 512
Wt = (torch.randn(N, K, device="cuda", generator=g) / math.sqrt(K)).to(torch.bfloat16)
for norm in (1, 0):
    for small in (1, 0):
        worst = {}
        for seed in range(40):
            row = torch.randn(1, K, device="cuda", generator=g) * (1 + seed)
            A = row.expand(M, K).contiguous()
            o = torch.zeros(M, N, device="cuda")
            _lib.check(lib.mt3_op_gemm(_lib.MT3_BF16, A.data_ptr(), 1, norm, Wt.data_ptr(), o.data_ptr(), M, N, K, _lib.EPI_F32, None, 0, small, s()))
            torch.cuda.synchronize()
            d = (o - o[0:1]).abs().max(1).values
            for r in d.nonzero().flatten().tolist():
                worst[r % 32] = max(worst.get(r % 32, 0), float(d[r]) / float(o[0].abs().max()))
        print(f"norm={norm} small={small}: rows (mod 32) that ever differ from row 0 (rel):", dict(sorted(worst.items())))
