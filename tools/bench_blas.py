#!/usr/bin/env python3
"""Vendor-library reference point: torch.matmul (hipBLASLt / rocBLAS) on the path's GEMM shapes (plain, no epilogue)."""
import torch
M = 40960
def timeit(fn, n=30):
    for _ in range(5): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
g = torch.Generator().manual_seed(0)
for name, N, K in [("qkv", 1152, 384), ("proj", 384, 384), ("fc1", 1536, 384), ("fc2", 384, 1536), ("dec fc1", 1024, 256), ("heads", 1664, 256)]:
    a = torch.randn(M, K, generator=g).cuda().bfloat16()
    w = torch.randn(N, K, generator=g).cuda().bfloat16()
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    us = timeit(lambda: torch.matmul(a, w.t(), out=out))
    dy = torch.randn(M, N, generator=g).cuda().bfloat16()
    dw = torch.empty(N, K, device="cuda", dtype=torch.bfloat16)
    us2 = timeit(lambda: torch.matmul(dy.t(), a, out=dw))
    print("%-8s N %5d K %5d   NT %7.1f us %7.1f TF   |  TN(wgrad) %7.1f us %7.1f TF" % (name, N, K, us, 2.0 * M * N * K / us / 1e6, us2, 2.0 * M * N * K / us2 / 1e6), flush=True)
