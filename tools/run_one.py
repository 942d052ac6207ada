#!/usr/bin/env python3
"""Run one kernel family a few times (for rocprofv3 --pmc passes).  usage: run_one.py attn_fwd|attn_bwd|gemm|wgrad [hd]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from star_amd import ops
what = sys.argv[1]
dev = "cuda"
g = torch.Generator().manual_seed(0)
if what.startswith("attn"):
    hd = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    NB, L, H = 40, 1024, 384 // hd if hd == 32 else 256 // hd
    qkv = torch.randn(NB * L, 3 * H * hd, generator=g).to(dev).bfloat16()
    dout = torch.randn(NB * L, H * hd, generator=g).to(dev).bfloat16()
    out, lse = ops.attn_fwd(qkv, NB, L, H, hd)
    for _ in range(3):
        if what == "attn_fwd":
            ops.attn_fwd(qkv, NB, L, H, hd)
        else:
            ops.attn_bwd(qkv, out, dout, lse, NB, L, H, hd)
elif what == "gemm":
    M, N, K = 40960, int(sys.argv[2]), int(sys.argv[3])
    a = torch.randn(M, K, generator=g).to(dev).bfloat16()
    w = torch.randn(N, K, generator=g).to(dev).bfloat16()
    for _ in range(3):
        ops.gemm(a, w)
elif what == "wgrad":
    M, N, K = 40960, int(sys.argv[2]), int(sys.argv[3])
    dy = torch.randn(M, N, generator=g).to(dev).bfloat16()
    x = torch.randn(M, K, generator=g).to(dev).bfloat16()
    for _ in range(3):
        ops.gemm_wgrad(dy, x)
elif what == "vq":
    K, E = int(sys.argv[2]), int(sys.argv[3])
    x = torch.relu(torch.randn(40960, E, generator=g)).to(dev)
    cb = (torch.randn(K, E, generator=g) * 0.5).abs().to(dev)
    for _ in range(3):
        ops.vq_assign(x, cb, 0)
torch.cuda.synchronize()
