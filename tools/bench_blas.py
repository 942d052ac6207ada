#!/usr/bin/env python3
"""Yardstick: the library GEMM (torch.matmul -> hipBLASLt / rocBLAS) at the shapes of the training step's plain Linears, beside star_gemm."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from star_amd import ops
from tools.bench_ops import timeit, M, dev

g = torch.Generator().manual_seed(0)
for name, N, K in [("qkv", 1152, 384), ("proj", 384, 384), ("fc1", 1536, 384), ("fc2 / dfc1", 384, 1536), ("dqkv", 384, 1152), ("dfc2", 1536, 384),
                   ("dec qkv", 768, 256), ("dec fc1", 1024, 256), ("dec fc2 / dfc1", 256, 1024), ("dec dqkv", 256, 768), ("heads", 1664, 256)]:
    a = torch.randn(M, K, generator=g).to(dev).bfloat16()
    w = (torch.randn(N, K, generator=g) * K ** -0.5).to(dev).bfloat16()
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    wt = w.t()
    us_t = timeit(lambda: torch.matmul(a, wt, out=out))
    us_s = timeit(lambda: ops.gemm(a, w, want_lp=True, out_lp=out))
    fl = 2.0 * M * N * K
    print("%-16s N %5d K %5d   library %7.1f us %7.1f TFLOP/s   star_gemm %7.1f us %7.1f TFLOP/s" % (name, N, K, us_t, fl / us_t / 1e6, us_s, fl / us_s / 1e6))
# weight gradients: dW = dY^T X
for name, N, K in [("w qkv", 1152, 384), ("w fc1", 1536, 384), ("w fc2", 384, 1536), ("w dec fc1", 1024, 256)]:
    dy = torch.randn(M, N, generator=g).to(dev).bfloat16()
    x = torch.randn(M, K, generator=g).to(dev).bfloat16()
    dyt = dy.t()
    us_t = timeit(lambda: torch.matmul(dyt, x))
    us_s = timeit(lambda: ops.gemm_wgrad_group([(dy, x, None, False)]))
    fl = 2.0 * M * N * K
    print("%-16s N %5d K %5d   library %7.1f us %7.1f TFLOP/s   grouped(1) %7.1f us %7.1f TFLOP/s" % (name, N, K, us_t, fl / us_t / 1e6, us_s, fl / us_s / 1e6))
