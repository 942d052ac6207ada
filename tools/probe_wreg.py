#!/usr/bin/env python3
"""A/B probe of the W-in-registers NT GEMM (STAR_AMD_GEMM_WREG=1) against torch on eligible shapes; prints time and max error."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from star_amd import ops  # noqa: E402

dev = "cuda"
g = torch.Generator().manual_seed(0)
print("WREG =", os.environ.get("STAR_AMD_GEMM_WREG", "0"), flush=True)
for M, N, K in ((192, 192, 384), (64 * 641, 384, 384), (40960, 1152, 384), (40960, 384, 384), (40960, 1536, 384)):
    a = torch.randn(M, K, generator=g).to(dev).bfloat16()
    w = (torch.randn(N, K, generator=g) * K ** -0.5).to(dev).bfloat16()
    bias = torch.randn(N, generator=g).to(dev)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    fn = lambda: ops.gemm(a, w, bias=bias, want_lp=True, want_f32=False, out_lp=out)  # noqa: E731
    fn()
    torch.cuda.synchronize()
    want = a.float() @ w.float().t() + bias
    err = (out.float() - want).abs().max().item()
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(30):
        fn()
    torch.cuda.synchronize()
    us = (time.perf_counter() - t) / 30 * 1e6
    print("M=%6d N=%5d K=%4d  %7.1f us  %6.1f TFLOP/s  max|err| %.4f" % (M, N, K, us, 2.0 * M * N * K / us / 1e6, err), flush=True)
