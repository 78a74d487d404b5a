"""Row-block permutation equivariance of the fused QKVG GEMM at the op level: the same rows in another order must give the same bits."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import gpu_util as U

def run(B, S=640, D=2048, K=2048, rope_heads=8, gate_act=1):
    M, N = B * S, 4 * D
    g = torch.Generator().manual_seed(3)
    A = (torch.rand((M + 256, K), generator=g) * 2 - 1).to(torch.bfloat16).cuda()
    W = ((torch.rand((N, K), generator=g) * 2 - 1) * 0.05).to(torch.bfloat16).cuda()
    qk_w = (1 + 0.1 * torch.randn((2 * D,), generator=g)).to(torch.bfloat16).cuda()
    ang = torch.rand((S, 64), generator=g).cuda()
    rope = torch.stack([torch.cos(ang), torch.sin(ang)], -1).contiguous()
    def go(Ain):
        C = torch.zeros((M, N), dtype=torch.bfloat16, device="cuda")
        S8 = (S + 7) // 8 * 8
        vt = torch.zeros((B, D, S8), dtype=torch.bfloat16, device="cuda")
        qkv = dict(D=D, S=S, rope_heads=rope_heads, pos0=0, eps=1e-6, qk_w=qk_w, rope=rope, vt=vt, vt_ld=S8, vt_row_stride=D * S8)
        U.gemm(Ain, W, C, cfg=5, M=M, N=N, K=K, lda=K, ldw=K, ldc=N, Npad=N, qkv=qkv)
        torch.cuda.synchronize()
        return C, vt
    C0, V0 = go(A)
    C0b, V0b = go(A)
    perm = torch.randperm(B, generator=g)
    Ap = A.clone()
    Ap[:M] = A[:M].view(B, S, K)[perm].reshape(M, K)
    C1, V1 = go(Ap)
    want = C0.view(B, S, N)[perm].reshape(M, N)
    bad = (C1 != want)
    print(f"B={B} S={S} rope_heads={rope_heads}: rerun equal {torch.equal(C0, C0b)}; permuted equal {not bool(bad.any())}; mismatching elements {int(bad.sum())}"
          + (f" in columns {sorted(set((bad.nonzero()[:, 1] // 128).tolist()))[:20]} rows%16 {sorted(set((bad.nonzero()[:, 0] % 64).tolist()))[:70]}" if bad.any() else ""), flush=True)

for (B, S) in ((24, 640), (3, 436), (5, 200), (24, 768), (7, 100), (2, 2560), (3, 37)):
    for rh in (8, 0):
        run(B, S=S, rope_heads=rh)
