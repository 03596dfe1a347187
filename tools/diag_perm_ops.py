import os, sys, math, ctypes as C
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mt3_amd import _lib
lib = _lib.load()
s = lambda: torch.cuda.current_stream().cuda_stream
g = torch.Generator(device="cuda").manual_seed(0)
BF = _lib.MT3_BF16
def gemm(A, a_f32, norm, Wt, out, M, N, K, epi, small):
    _lib.check(lib.mt3_op_gemm(BF, A.data_ptr(), int(a_f32), int(norm), Wt.data_ptr(), out.data_ptr(), M, N, K, epi, None, 0, int(small), s()))
    torch.cuda.synchronize()
for M in (64, 256):
    perm = torch.randperm(M, device="cuda")
    for small in (True, False):
        for (name, a_f32, norm, epi, N, K, odt) in [("QKV norm store", 1, 1, _lib.EPI_STORE, 1152, 512, torch.bfloat16),
                                                     ("resid", 0, 0, _lib.EPI_RESID, 512, 384, torch.float32),
                                                     ("resid K1024", 0, 0, _lib.EPI_RESID, 512, 1024, torch.float32),
                                                     ("geglu", 1, 1, _lib.EPI_GEGLU, 2048, 512, torch.bfloat16),
                                                     ("logits f32", 1, 1, _lib.EPI_F32, 1536, 512, torch.float32)]:
            A = torch.randn(M, K, device="cuda", generator=g)
            if not a_f32: A = A.to(torch.bfloat16)
            Wt = (torch.randn(N, K, device="cuda", generator=g) / math.sqrt(K)).to(torch.bfloat16)
            No = N // 2 if epi == _lib.EPI_GEGLU else N
            base = torch.randn(M, No, device="cuda", generator=g).to(odt)
            o1 = base.clone(); gemm(A, a_f32, norm, Wt, o1, M, N, K, epi, small)
            o2 = base[perm].clone(); gemm(A[perm].contiguous(), a_f32, norm, Wt, o2, M, N, K, epi, small)
            d = (o2.float() - o1[perm].float()).abs().max().item()
            print(f"M={M} small={small} {name}: perm diff {d:.3e}")
# decode attention
B, H, cap = 256, 6, 1024
perm = torch.randperm(B, device="cuda")
kc = torch.randn(B, H, cap, 64, device="cuda", generator=g).to(torch.bfloat16)
vc = torch.randn(B, H, cap, 64, device="cuda", generator=g).to(torch.bfloat16)
q = (torch.randn(B, H * 64, device="cuda", generator=g) * 0.3).to(torch.bfloat16)
for n in (1, 37, 256, 1000):
    o1 = torch.zeros(B, H * 64, device="cuda", dtype=torch.bfloat16); o2 = o1.clone()
    _lib.check(lib.mt3_op_decode_attention(BF, q.data_ptr(), H * 64, kc.data_ptr(), vc.data_ptr(), cap, None, None, 0, None, n, o1.data_ptr(), B, H, s()))
    kc2, vc2, q2 = kc[perm].contiguous(), vc[perm].contiguous(), q[perm].contiguous()
    _lib.check(lib.mt3_op_decode_attention(BF, q2.data_ptr(), H * 64, kc2.data_ptr(), vc2.data_ptr(), cap, None, None, 0, None, n, o2.data_ptr(), B, H, s()))
    torch.cuda.synchronize()
    print(f"dec_attn n_keys={n}: perm diff {(o2.float() - o1[perm].float()).abs().max().item():.3e}")
