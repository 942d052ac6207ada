#!/usr/bin/env python3
"""Forward/backward attention time at a fixed number of scores, varying the sequence length (block-overhead probe)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from star_amd import ops
g = torch.Generator().manual_seed(0)
for hd, H in ((32, 12), (16, 16)):
    for NB, L in ((160, 512), (40, 1024), (10, 2048), (2, 4096 + 512)):
        qkv = torch.randn(NB * L, 3 * H * hd, generator=g).cuda().bfloat16()
        dout = torch.randn(NB * L, H * hd, generator=g).cuda().bfloat16()
        out, lse = ops.attn_fwd(qkv, NB, L, H, hd)
        def timeit(fn, n=20):
            for _ in range(3): fn()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(n): fn()
            b.record(); torch.cuda.synchronize()
            return a.elapsed_time(b) / n * 1e3
        tf = timeit(lambda: ops.attn_fwd(qkv, NB, L, H, hd))
        tb = timeit(lambda: ops.attn_bwd(qkv, out, dout, lse, NB, L, H, hd))
        sc = NB * H * L * L / 1e6
        print("hd %2d NB %4d L %5d  Mscores %7.1f  fwd %7.1f us (%.3f ns/kscore)  bwd %7.1f us" % (hd, NB, L, sc, tf, tf * 1e3 / sc / 1e3, tb), flush=True)
