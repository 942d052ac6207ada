#!/usr/bin/env python3
"""Micro-benchmarks of the kernel families at the shapes of the B=8 training step (GPU box only)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from star_amd import ops
from star_amd._lib import ACT_GELU, ACT_DGELU, ACT_MULAUX, ACT_NONE, GEMM_PRE_IS_GRAD

dev = "cuda"
M = int(os.environ.get("M", 40960))


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3  # us


def gemm_cases():
    g = torch.Generator().manual_seed(0)
    rows = []
    for name, N, K, kw in [
        ("qkv      ", 1152, 384, dict(bias=True)),
        ("proj+res ", 384, 384, dict(bias=True, res=True)),
        ("fc1+gelu ", 1536, 384, dict(bias=True, act=ACT_GELU, pre=True, flags=GEMM_PRE_IS_GRAD)),
        ("fc2+res  ", 384, 1536, dict(bias=True, res=True)),
        ("dfc2 mula", 1536, 384, dict(act=ACT_MULAUX, aux=True)),
        ("dfc1     ", 384, 1536, dict()),
        ("dqkv     ", 384, 1152, dict()),
        ("dec qkv  ", 768, 256, dict(bias=True)),
        ("dec fc1  ", 1024, 256, dict(bias=True, act=ACT_GELU, pre=True, flags=GEMM_PRE_IS_GRAD)),
        ("dec fc2  ", 256, 1024, dict(bias=True, res=True)),
        ("heads    ", 1664, 256, dict(bias=True, f32=True)),
    ]:
        a = torch.randn(M, K, generator=g).to(dev).bfloat16()
        w = (torch.randn(N, K, generator=g) * K ** -0.5).to(dev).bfloat16()
        bias = torch.randn(N, generator=g).to(dev) if kw.get("bias") else None
        res = torch.randn(M, N, generator=g).to(dev) if kw.get("res") else None
        aux = torch.randn(M, N, generator=g).to(dev).bfloat16() if kw.get("aux") else None
        f32 = bool(kw.get("res") or kw.get("f32"))
        out32 = torch.empty(M, N, device=dev) if f32 else None
        outlp = None if f32 else torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        fn = lambda: ops.gemm(a, w, bias=bias, act=kw.get("act", ACT_NONE), res=res, aux=aux, want_pre=bool(kw.get("pre")),
                              want_lp=not f32, want_f32=f32, out_f32=out32, out_lp=outlp, flags=kw.get("flags", 0))
        us = timeit(fn)
        fl = 2.0 * M * N * K
        byt = M * K * 2 + M * N * ((4 if f32 else 2) + (2 if kw.get("pre") else 0) + (4 if kw.get("res") else 0) + (2 if kw.get("aux") else 0))
        rows.append((name, N, K, us, fl / us / 1e6, byt / us / 1e3))
    return rows


def wgrad_cases():
    g = torch.Generator().manual_seed(0)
    rows = []
    for name, N, K in [("w qkv", 1152, 384), ("w proj", 384, 384), ("w fc1", 1536, 384), ("w fc2", 384, 1536), ("w dqkv", 768, 256),
                       ("w dfc1", 1024, 256), ("w dfc2", 256, 1024), ("w heads", 1664, 256)]:
        dy = torch.randn(M, N, generator=g).to(dev).bfloat16()
        x = torch.randn(M, K, generator=g).to(dev).bfloat16()
        dw = torch.empty(N, K, device=dev)
        db = torch.empty(N, device=dev)
        us = timeit(lambda: ops.gemm_wgrad(dy, x, dw=dw, db=db))
        rows.append((name, N, K, us, 2.0 * M * N * K / us / 1e6, (M * (N + K) * 2) / us / 1e3))
    return rows


def conv_cases():
    """the ConvPred head's three 3 x 3 convolutions at 20 frames (joint model, B = 4): forward, data gradient, weight gradient."""
    g = torch.Generator().manual_seed(0)
    rows = []
    n = int(os.environ.get("FRAMES", 20))
    for name, h, cin, cout in [("conv1 64^2 ", 64, 512, 256), ("conv2 128^2", 128, 256, 128), ("conv3 256^2", 256, 128, 32)]:
        R = n * h * h
        x = torch.randn(R, cin, generator=g).to(dev).bfloat16()
        dy = torch.randn(R, cout, generator=g).to(dev).bfloat16()
        wf = (torch.randn(cout, 9 * cin, generator=g) * (9 * cin) ** -0.5).to(dev).bfloat16()
        wd = (torch.randn(cin, 9 * cout, generator=g) * (9 * cout) ** -0.5).to(dev).bfloat16()
        out = torch.empty(R, cout, device=dev)
        dx = torch.empty(R, cin, device=dev)
        fl = 2.0 * R * 9 * cin * cout
        us = timeit(lambda: ops.gemm(x, wf, want_lp=False, want_f32=True, out_f32=out, conv=(h, h)), n=5)
        rows.append((name + " fwd", cout, 9 * cin, us, fl / us / 1e6, (R * cin * 2 + R * cout * 4) / us / 1e3))
        us = timeit(lambda: ops.gemm(dy, wd, want_lp=False, want_f32=True, out_f32=dx, conv=(h, h)), n=5)
        rows.append((name + " dgrad", cin, 9 * cout, us, fl / us / 1e6, (R * cout * 2 + R * cin * 4) / us / 1e3))
        dwt = torch.empty(9, cout, cin, device=dev)
        us = timeit(lambda: ops.gemm_wgrad_group([(dy, x, dwt[t], False, ops.wgrad_tap(t // 3 - 1, t % 3 - 1, h, h)) for t in range(9)]), n=5)
        rows.append((name + " wgrad", cout, cin, us, fl / us / 1e6, (R * cout * 2 + R * cin * 2) / us / 1e3))
    return rows


def wgroup_cases():
    """a block's four weight gradients: one grouped launch against the four separate ones."""
    g = torch.Generator().manual_seed(0)
    rows = []
    for name, shapes in [("wg enc block", [(1152, 384), (384, 384), (1536, 384), (384, 1536)]),
                         ("wg dec block", [(768, 256), (256, 256), (1024, 256), (256, 1024)]),
                         ("wg heads+dec", [(1664, 256), (768, 256), (256, 256), (1024, 256), (256, 1024)])]:
        jobs = []
        for N, K in shapes:
            jobs.append((torch.randn(M, N, generator=g).to(dev).bfloat16(), torch.randn(M, K, generator=g).to(dev).bfloat16(),
                         torch.empty(N, K, device=dev), torch.empty(N, device=dev)))
        fl = sum(2.0 * M * N * K for N, K in shapes)
        byt = sum(M * (N + K) * 2 for N, K in shapes)
        us = timeit(lambda: ops.gemm_wgrad_group(jobs))
        rows.append((name + " grouped", 0, 0, us, fl / us / 1e6, byt / us / 1e3))
        us = timeit(lambda: [ops.gemm_wgrad(dy, x, dw=dw, db=db) for dy, x, dw, db in jobs])
        rows.append((name + " separate", 0, 0, us, fl / us / 1e6, byt / us / 1e3))
    return rows


def gemm_ln_cases():
    """residual Linear + the LayerNorm behind it: one kernel against the two launches."""
    rows = []
    g = torch.Generator().manual_seed(0)
    for name, N, K in [("proj+res+ln", 384, 384), ("fc2+res+ln", 384, 1536), ("dec proj+ln", 256, 256), ("dec fc2+ln", 256, 1024)]:
        a = torch.randn(M, K, generator=g).to(dev).bfloat16()
        w = (torch.randn(N, K, generator=g) * K ** -0.5).to(dev).bfloat16()
        bias = torch.randn(N, generator=g).to(dev)
        res = torch.randn(M, N, generator=g).to(dev)
        gam, bet = torch.ones(N, device=dev), torch.zeros(N, device=dev)
        out = torch.empty(M, N, device=dev)

        def two():
            ops.gemm(a, w, bias=bias, res=res, want_lp=False, want_f32=True, out_f32=out)
            ops.layernorm_fwd(out, gam, bet, 1e-6, 0)
        us2 = timeit(two)
        us1 = timeit(lambda: ops.gemm_res_ln(a, w, bias, res, gam, bet, 1e-6, out=out))
        byt = M * K * 2 + M * N * 10
        rows.append((name + " 2 launches", N, K, us2, 2.0 * M * N * K / us2 / 1e6, byt / us2 / 1e3))
        rows.append((name + " fused", N, K, us1, 2.0 * M * N * K / us1 / 1e6, byt / us1 / 1e3))
    return rows


def attn_cases():
    rows = []
    g = torch.Generator().manual_seed(0)
    for name, NB, L, H, hd in [("enc attn", M // 1024, 1024, 12, 32), ("dec attn", M // 1024, 1024, 16, 16)]:
        qkv = torch.randn(NB * L, 3 * H * hd, generator=g).to(dev).bfloat16()
        dout = torch.randn(NB * L, H * hd, generator=g).to(dev).bfloat16()
        out, lse = ops.attn_fwd(qkv, NB, L, H, hd)
        us = timeit(lambda: ops.attn_fwd(qkv, NB, L, H, hd))
        rows.append((name + " fwd", H, hd, us, 4.0 * NB * H * L * L * hd / us / 1e6, NB * H * L * L / us / 1e3))
        us = timeit(lambda: ops.attn_bwd(qkv, out, dout, lse, NB, L, H, hd))
        rows.append((name + " bwd", H, hd, us, 10.0 * NB * H * L * L * hd / us / 1e6, 2 * NB * H * L * L / us / 1e3))
    return rows


def ln_cases():
    rows = []
    g = torch.Generator().manual_seed(0)
    for D in (384, 256):
        x = torch.randn(M, D, generator=g).to(dev)
        gam, bet = torch.ones(D, device=dev), torch.zeros(D, device=dev)
        y, mean, rstd = ops.layernorm_fwd(x, gam, bet, 1e-6, 0)
        us = timeit(lambda: ops.layernorm_fwd(x, gam, bet, 1e-6, 0))
        rows.append(("ln fwd", D, 0, us, 0.0, M * D * 6 / us / 1e3))
        dy = torch.randn(M, D, generator=g).to(dev).bfloat16()
        gin = torch.randn(M, D, generator=g).to(dev)
        gout = torch.empty(M, D, device=dev)
        us = timeit(lambda: ops.layernorm_bwd(dy, x, gam, mean, rstd, g_in=gin, g_out=gout))
        rows.append(("ln bwd", D, 0, us, 0.0, M * D * 16 / us / 1e3))
    return rows


def voxel_cases():
    """Batch of B=8 scenes x 5 agents x (T + teacher) frames, 3 % occupancy: one scatter launch."""
    import numpy as np
    rows = []
    rng = np.random.RandomState(0)
    for F in (40, 80, 160):
        n_per = int(0.03 * 256 * 256 * 13)
        idx = torch.from_numpy(np.stack([rng.randint(0, d, size=F * n_per) for d in (256, 256, 13)], 1).astype(np.int32)).to(dev)
        off = (torch.arange(F + 1, dtype=torch.int64) * n_per).to(dev)
        out = torch.empty(F, 256, 256, 13, device=dev)
        hoff = off.cpu()
        us = timeit(lambda: ops.voxel_scatter(idx, off, (256, 256, 13), validate=False, out=out, host_offsets=hoff))
        byt = F * 256 * 256 * 13 * 4 + F * n_per * 16
        rows.append(("voxel %d fr" % F, F, n_per, us, 0.0, byt / us / 1e3))
    return rows


def vq_cases():
    rows = []
    g = torch.Generator().manual_seed(0)
    for K, E in ((8192, 32), (512, 64)):
        x = torch.relu(torch.randn(M, E, generator=g)).to(dev)
        cb = (torch.randn(K, E, generator=g) * 0.5).abs().to(dev)
        v = ops.vq_assign(x, cb, 0)
        us = timeit(lambda: ops.vq_assign(x, cb, 0))
        rows.append(("vq assign", K, E, us, 2.0 * M * K * E / us / 1e6, (M * E * 10 + K * E * 4) / us / 1e3))
        dq = torch.randn(M, E, generator=g).to(dev).bfloat16()
        us = timeit(lambda: ops.vq_bwd(dq, x, v["idx"], cb, 0.25))
        rows.append(("vq bwd", K, E, us, 0.0, (M * E * 8 + K * E * 8) / us / 1e3))
        # collapsed codebook: every row picks code 0 (what early VQ-STAR training looks like)
        cb2 = cb + 100.0
        cb2[0] = x.mean(0)
        v2 = ops.vq_assign(x, cb2, 0)
        assert int(v2["counts"].max()) == M
        us = timeit(lambda: ops.vq_assign(x, cb2, 0))
        rows.append(("vq assign 1code", K, E, us, 2.0 * M * K * E / us / 1e6, 0.0))
        us = timeit(lambda: ops.vq_bwd(dq, x, v2["idx"], cb2, 0.25))
        rows.append(("vq bwd 1code", K, E, us, 0.0, 0.0))
    return rows


if __name__ == "__main__":
    which = sys.argv[1:] or ["gemm", "wgrad", "wgroup", "attn", "ln", "voxel", "vq"]
    for w in which:
        rows = dict(gemm=gemm_cases, gemm_ln=gemm_ln_cases, wgrad=wgrad_cases, wgroup=wgroup_cases, conv=conv_cases, attn=attn_cases, ln=ln_cases, voxel=voxel_cases, vq=vq_cases)[w]()
        for name, a, b, us, tf, gbs in rows:
            print("%-26s %5d %5d  %8.1f us  %7.1f TFLOP/s  %7.1f GB/s(or Gexp/s)" % (name, a, b, us, tf, gbs))
