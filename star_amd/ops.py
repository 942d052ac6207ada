"""Tensor-level wrappers over the C ABI (one per kernel group).  PyTorch only allocates and hands over
device pointers here; all arithmetic happens in libstar_hip.so."""
import ctypes as C
import math

import numpy as np
import os

import torch

from . import _lib as L
from ._lib import ACT_DGELU, ACT_DRELU, ACT_GELU, ACT_MULAUX, ACT_NONE, ACT_RELU, BF16, BF16X3, F32, GEMM_A_PLAIN, p  # noqa: F401


class Profiler:
    """Optional per-launch HIP-event timing of the C-ABI calls (bench.py's roofline leg).  Events are
    recorded on torch's current stream, which is the stream every kernel here is launched on."""

    def __init__(self, tags=None):
        self.tags, self.records = tags, []
        self.active = True      # bench.py samples every few steps: an event pair per launch costs about 1 % of a step per 15 launches

    def want(self, tag):
        return self.active and (self.tags is None or tag in self.tags)

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for tag, flops, nbytes, e0, e1 in self.records:
            d = out.setdefault(tag, dict(launches=0, ms=0.0, flops=0.0, bytes=0.0))
            d["launches"] += 1
            d["ms"] += e0.elapsed_time(e1)
            d["flops"] += flops
            d["bytes"] += nbytes
        return out


PROFILER = None


class _timed:
    def __init__(self, tag, flops=0.0, nbytes=0.0):
        self.on = PROFILER is not None and PROFILER.want(tag)
        self.tag, self.flops, self.nbytes = tag, flops, nbytes

    def __enter__(self):
        if self.on:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e1 = torch.cuda.Event(enable_timing=True)
            self.e0.record()

    def __exit__(self, *exc):
        if self.on:
            self.e1.record()
            PROFILER.records.append((self.tag, self.flops, self.nbytes, self.e0, self.e1))


def _cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("star_amd ops need device tensors (got a CPU tensor)")


# ---------------------------------------------------------------------------------------------------
def warp_thetas(trans_matrices):
    """Host part of feature_transformation (reference star/models/mae_base.py:593-606): the rotation
    block and the normalised translation of every (scene, ego, neighbour) pose, formed in the SOURCE
    dtype of `trans_matrices` and only then cast to float32.  Returns float32 [B, A, A, 8] (CPU)."""
    t = trans_matrices.detach().cpu() if isinstance(trans_matrices, torch.Tensor) else torch.as_tensor(np.asarray(trans_matrices))
    out = torch.zeros(t.shape[:3] + (8,), dtype=torch.float32)
    out[..., 0] = t[..., 0, 0].to(torch.float32)
    out[..., 1] = t[..., 0, 1].to(torch.float32)
    out[..., 2] = t[..., 1, 0].to(torch.float32)
    out[..., 3] = t[..., 1, 1].to(torch.float32)
    out[..., 4] = ((4 * t[..., 0, 3]) / 128).to(torch.float32)
    out[..., 5] = (-(4 * t[..., 1, 3]) / 128).to(torch.float32)
    return out


def warp_fuse(pred_u8, theta, num_agent, batch, ego=None, binary=False):
    """K10.  pred_u8 uint8 [A*B,C,H,W]; theta float32 [B,A,A,8] on the device; num_agent int [B,A] on
    the device; ego optional float32 [A*B,C,H,W] with any strides.  -> float32 [A*B,C,H,W].
    binary=True promises {0,1} maps and takes the bit-packed fast path (identical result)."""
    _cuda(pred_u8, theta, num_agent, ego)
    ab, c, h, w = pred_u8.shape
    assert ab % batch == 0, (pred_u8.shape, batch)
    a = ab // batch
    assert pred_u8.dtype == torch.uint8 and pred_u8.is_contiguous()
    assert theta.dtype == torch.float32 and theta.is_contiguous() and tuple(theta.shape) == (batch, a, a, 8)
    assert num_agent.dtype in (torch.int32, torch.int64) and num_agent.is_contiguous() and tuple(num_agent.shape) == (batch, a)
    out = torch.empty(ab, c, h, w, dtype=torch.float32, device=pred_u8.device)
    es = (0, 0, 0, 0)
    if ego is not None:
        assert ego.dtype == torch.float32 and tuple(ego.shape) == (ab, c, h, w)
        es = ego.stride()
    ws = None
    if binary and c <= 16:
        ws = torch.empty(L.lib().star_warp_fuse_ws(batch, a, h, w) // 8 + 1, dtype=torch.int64, device=pred_u8.device)
    # bytes the launch has to move in ITS operand formats.  Bit-packed path (what training and inference run): the uint8 decisions
    # once (pack), one 2-byte word per pixel written by the pack and gathered per (pixel, agent) by the fusion, the float32 map
    # written once, the float32 ego map read once in inference.  (SURVEY 8d's (A + 1) x 3.41 MB per ego frame prices the reference's
    # float32 gathers: 4-5x these bytes.)  Byte-gather path: A uint8 gathers per voxel instead.
    hw = float(h) * w
    if ws is not None:
        nbytes = ab * (c * hw * 1 + hw * 2 + a * hw * 2 + c * hw * 4 + (c * hw * 4 if ego is not None else 0))
    else:
        nbytes = ab * (a * c * hw * 1 + c * hw * 4 + (c * hw * 4 if ego is not None else 0))
    with _timed("warp_fuse", 0.0, float(nbytes)):
        L.check(L.lib().star_warp_fuse(p(pred_u8), p(ego), es[0], es[1], es[2], es[3], p(theta), p(num_agent),
                                       int(num_agent.dtype == torch.int64), p(out), p(ws), batch, a, c, h, w, L.stream()))
    return out


def mask_sort(noise):
    """K5.  noise float32 [R, L] -> (ids_shuffle, ids_restore) int64 [R, L]."""
    _cuda(noise)
    assert noise.dtype == torch.float32 and noise.dim() == 2
    noise = noise.contiguous()
    r, l = noise.shape
    ids_shuffle = torch.empty(r, l, dtype=torch.int64, device=noise.device)
    ids_restore = torch.empty(r, l, dtype=torch.int64, device=noise.device)
    L.check(L.lib().star_mask_sort(p(noise), p(ids_shuffle), p(ids_restore), r, l, L.stream()))
    return ids_shuffle, ids_restore


def build_rowmaps(ids_shuffle, ids_restore, ba, t, l, keep, random_mode, single, want_mask=True):
    dev = ids_shuffle.device
    rows = ba * t * keep
    src = torch.empty(rows, dtype=torch.int32, device=dev)
    dst = torch.empty(rows, dtype=torch.int32, device=dev)
    mask = torch.empty((ba * t, l) if random_mode else (ba, t, l), dtype=torch.float32, device=dev) if want_mask else None
    fill = torch.empty(ba * l, dtype=torch.int32, device=dev)
    L.check(L.lib().star_build_rowmaps(p(ids_shuffle), p(ids_restore), p(src), p(dst), p(mask), p(fill), ba, t, l,
                                       keep, int(random_mode), int(single), L.stream()))
    return src, dst, mask, fill


# ---------------------------------------------------------------------------------------------------
def layernorm_fwd(x, gamma, beta, eps, dt, want_stats=True):
    _cuda(x, gamma, beta)
    m, d = x.shape
    assert x.dtype == torch.float32 and x.is_contiguous()
    y = torch.empty(m, d * L.width(dt), dtype=L.torch_dtype(dt), device=x.device)       # BF16X3: rows of [hi | lo]
    mean = torch.empty(m, dtype=torch.float32, device=x.device) if want_stats else None
    rstd = torch.empty(m, dtype=torch.float32, device=x.device) if want_stats else None
    with _timed("layernorm_fwd", 0.0, float(m) * d * (4 + (2 if dt == BF16 else 4))):
        L.check(L.lib().star_layernorm_fwd(p(x), p(gamma), p(beta), p(y), p(mean), p(rstd), m, d, float(eps), dt, L.stream()))
    return y, mean, rstd


def layernorm_bwd(dy, x, gamma, mean, rstd, g_in=None, want_lp=True, g_out=None, dgamma=None, dbeta=None, defer_ws=None):
    """-> (g_out float32 [M,D] = g_in + dx, g_lp copy in dy's dtype or None, dgamma, dbeta).  defer_ws: a caller-kept workspace;
    the partial dgamma / dbeta rows are left in it for ln_fold_batch and dgamma / dbeta are NOT written by this call."""
    m, d = x.shape
    dt = L.dt_of(dy)
    assert dy.is_contiguous() and x.is_contiguous()
    if g_out is None:
        g_out = torch.empty(m, d, dtype=torch.float32, device=x.device)
    g_lp = torch.empty(m, d, dtype=dy.dtype, device=x.device) if want_lp else None
    if dgamma is None:
        dgamma = torch.empty(d, dtype=torch.float32, device=x.device)
    if dbeta is None:
        dbeta = torch.empty(d, dtype=torch.float32, device=x.device)
    defer = defer_ws is not None
    ws = defer_ws if defer else torch.empty(L.lib().star_layernorm_bwd_ws(d), dtype=torch.float32, device=x.device)
    assert ws.numel() >= L.lib().star_layernorm_bwd_ws(d) and ws.dtype == torch.float32
    with _timed("layernorm_bwd", 0.0, float(m) * d * (4 + 4 + 4 + 2 * (2 if dt == BF16 else 4))):
        L.check(L.lib().star_layernorm_bwd(p(dy), p(x), p(gamma), p(mean), p(rstd), p(g_in), p(g_out), p(g_lp),
                                           None if defer else p(dgamma), None if defer else p(dbeta), p(ws), m, d, dt, L.stream()))
    return g_out, g_lp, dgamma, dbeta


def ln_fold_table(jobs, device):
    """Device job table for ln_fold_batch: jobs = [(ws, dgamma, dbeta, M, D), ...] -> (table tensor, njobs, max D)."""
    rec = np.zeros(len(jobs), dtype=np.dtype([("ws", "<u8"), ("dg", "<u8"), ("db", "<u8"), ("nb", "<i4"), ("D", "<i4")]))
    for i, (ws, dg, db, m, d) in enumerate(jobs):
        assert dg.is_contiguous() and db.is_contiguous() and dg.numel() == d and db.numel() == d
        rec[i] = (ws.data_ptr(), dg.data_ptr(), db.data_ptr(), L.lib().star_layernorm_fold_rows(m), d)
    tab = torch.from_numpy(rec.view(np.uint8).copy()).to(device)
    return tab, len(jobs), max(j[4] for j in jobs)


def ln_fold_batch(tab, njobs, max_d):
    L.check(L.lib().star_layernorm_fold_batch(p(tab), njobs, max_d, L.stream()))


# ---------------------------------------------------------------------------------------------------
def gemm(a, w, m=None, bias=None, act=ACT_NONE, res=None, res_mod=0, res2=None, res2_div=0, res2_mod=0, aux=None,
         a_rows=None, c_rows=None, out_rows=None, want_pre=False, want_lp=True, want_f32=False, out_f32=None, out_lp=None,
         flags=0, dt=None, conv=None):
    """C = epilogue(A[a_rows] @ W^T).  a [*, K], w [N, K] same dtype (bf16 or f32).
    conv = (h, w): a [pixels, Cin] channels-last images, w [N, 9 Cin] tap-major: the 3 x 3 convolution (padding 1) as one product.
    dt = BF16X3: w [N, 2K] and a [*, 2K] are bf16 hi/lo pairs (a [*, K] plain bf16 with flags & GEMM_A_PLAIN), out_lp [rows, 2N].
    Returns (out_lp, out_f32, pre) -- entries not requested are None."""
    _cuda(a, w)
    if dt is None:
        dt = L.dt_of(a)
    assert w.dtype == a.dtype and a.is_contiguous() and w.is_contiguous()
    wd = L.width(dt)
    k = w.shape[1] // wd
    n = w.shape[0]
    cw = 0
    if conv is not None:
        ch, cwid = conv
        assert ch & (ch - 1) == 0 and cwid & (cwid - 1) == 0 and a.shape[-1] * 9 == k and a_rows is None, (conv, a.shape, w.shape)
        cw = 0x40000000 | (cwid.bit_length() - 1) | ((ch.bit_length() - 1) << 5)
    else:
        assert a.shape[-1] == (k if (dt != BF16X3 or flags & GEMM_A_PLAIN) else 2 * k), (a.shape, w.shape, dt, flags)
    if m is None:
        m = a_rows.numel() if a_rows is not None else a.shape[0]
    rows_out = out_rows if out_rows is not None else m
    dev = a.device
    if want_lp and out_lp is None:
        out_lp = (torch.zeros if c_rows is not None else torch.empty)(rows_out, n * wd, dtype=a.dtype, device=dev)
    if want_f32 and out_f32 is None:
        out_f32 = (torch.zeros if c_rows is not None else torch.empty)(rows_out, n, dtype=torch.float32, device=dev)
    assert not (want_pre and dt == BF16X3), "bf16x3 is a forward-only storage type"
    pre = torch.empty(rows_out, n, dtype=a.dtype, device=dev) if want_pre else None
    g = L.StarGemm(p(a), p(w), p(a_rows), p(c_rows), p(bias), p(res), res_mod, p(res2), res2_div, res2_mod, p(aux), act,
                   p(pre), p(out_lp), p(out_f32), m, n, k, dt, flags, cw)
    esz = a.element_size() * wd
    nbytes = (m * (k // 9 if conv is not None else k) + n * k) * esz + rows_out * n * ((esz if out_lp is not None else 0) + (4 if out_f32 is not None else 0) +
                                                      (esz if pre is not None else 0)) + m * n * ((4 if res is not None else 0) +
                                                                                                 (esz if aux is not None else 0))
    with _timed("gemm_nt", 2.0 * m * n * k, float(nbytes)):   # algorithmic bytes: every operand and result once
        L.check(L.lib().star_gemm(C.byref(g), L.stream()))
    return out_lp, out_f32, pre


def gemm_wgrad(dy, x, x_rows=None, want_bias=True, want_weight=True, dw=None, db=None):
    """dW [N,K] = dY^T X[x_rows], db [N] = colsum(dY)."""
    _cuda(dy, x)
    dt = L.dt_of(dy)
    assert x.dtype == dy.dtype and dy.is_contiguous() and x.is_contiguous()
    m, n = dy.shape
    k = x.shape[1]
    dev = dy.device
    if dw is None and want_weight:
        dw = torch.empty(n, k, dtype=torch.float32, device=dev)
    if db is None and want_bias:
        db = torch.empty(n, dtype=torch.float32, device=dev)
    assert dw is None or (dw.is_contiguous() and dw.numel() == n * k)
    assert db is None or (db.is_contiguous() and db.numel() == n)
    ws = torch.empty(L.lib().star_gemm_wgrad_ws(m, n, k), dtype=torch.float32, device=dev)
    with _timed("gemm_wgrad", 2.0 * m * n * k, float(m * (n + k) * dy.element_size() + n * k * 4)):
        L.check(L.lib().star_gemm_wgrad(p(dy), p(x), p(x_rows), p(dw), p(db), p(ws), m, n, k, dt, L.stream()))
    return dw, db


_WG_GROUP_WS = {}      # (device, stream) -> zero-initialised workspace of the grouped launches (ticket counters return to zero)


def wgrad_group_eligible(dy, x, x_rows=None):
    """can this product ride in a grouped launch (star_gemm_wgrad_group)?"""
    m, n = dy.shape
    k = x.shape[1]
    return (dy.dtype == torch.bfloat16 and x_rows is None and n % 8 == 0 and k % 8 == 0 and m * max(n, k) * 2 < 0x7fffffff and
            dy.data_ptr() % 16 == 0 and x.data_ptr() % 16 == 0)


def wgrad_tap(dy, dx, h, w):
    """STAR_WGRAD_TAP of star_hip.h; h, w powers of two."""
    assert h & (h - 1) == 0 and w & (w - 1) == 0 and -1 <= dy <= 1 and -1 <= dx <= 1
    return 0x40000000 | (dy + 1) | ((dx + 1) << 2) | ((w.bit_length() - 1) << 4) | ((h.bit_length() - 1) << 9)


def gemm_wgrad_group(jobs):
    """jobs: [(dy [M,N] bf16, x [M,K] bf16, dw float32 [N,K] or None, db float32 [N] or None / False = no bias[, tap]), ...] -> [(dw, db), ...].
    One launch for all of them (star_gemm_wgrad_group); more than WGRAD_GROUP_MAX jobs go out in several launches.
    tap = wgrad_tap(dy, dx, h, w): the product of one tap of a 3 x 3 convolution over channels-last h x w images (x rows shifted by
    (dy, dx) inside their image, zeros outside)."""
    out = []
    for lo in range(0, len(jobs), L.WGRAD_GROUP_MAX):
        part = jobs[lo:lo + L.WGRAD_GROUP_MAX]
        arr = (L.StarWgradJob * len(part))()
        flops = nbytes = 0.0
        dev = part[0][0].device
        res = []
        for i, job in enumerate(part):
            dy, x, dw, db = job[:4]
            tap = job[4] if len(job) > 4 else 0
            _cuda(dy, x)
            assert dy.dtype == torch.bfloat16 and x.dtype == dy.dtype and dy.is_contiguous() and x.is_contiguous() and dy.shape[0] == x.shape[0]
            m, n = dy.shape
            k = x.shape[1]
            if dw is None:
                dw = torch.empty(n, k, dtype=torch.float32, device=dev)
            if db is None:
                db = torch.empty(n, dtype=torch.float32, device=dev)
            elif db is False:
                db = None
            assert dw.is_contiguous() and dw.numel() == n * k and dw.dtype == torch.float32
            assert db is None or (db.is_contiguous() and db.numel() == n and db.dtype == torch.float32)
            arr[i] = L.StarWgradJob(p(dy), p(x), p(dw), p(db), m, n, k, tap)
            flops += 2.0 * m * n * k
            nbytes += float(m * (n + k) * 2 + n * k * 4)
            res.append((dw, db))
        need = L.lib().star_gemm_wgrad_group_ws(arr, len(part), BF16)
        if need < 0:
            raise RuntimeError("libstar_hip: " + L.lib().star_last_error().decode())
        key = (dev, L.stream())
        ws = _WG_GROUP_WS.get(key)
        if ws is None or ws.numel() < need:
            ws = _WG_GROUP_WS[key] = torch.zeros(max(need, 1 << 22), dtype=torch.float32, device=dev)
        with _timed("gemm_wgrad", flops, nbytes):
            L.check(L.lib().star_gemm_wgrad_group(arr, len(part), p(ws), ws.numel(), BF16, L.stream()))
        out += res
    return out


def gemm_res_ln_supported(n, k, dt):
    return bool(L.lib().star_gemm_res_ln_supported(n, k, dt))


def gemm_res_ln_prefers(m, n, k, dt, *operands):
    """a kernel exists AND is expected to beat star_gemm + star_layernorm_fwd at this M (whole-row tiles: M sets the tile count) AND
    every operand given (bias / gamma / beta / residual views) is 16-byte aligned, as star_gemm_res_ln requires -- parameters that
    are views at odd offsets into other storage take the two-launch path instead of raising"""
    if any(t is not None and t.data_ptr() % 16 != 0 for t in operands):
        return False
    return bool(L.lib().star_gemm_res_ln_prefers(m, n, k, dt))


def gemm_res_ln(a, w, bias, res, gamma, beta, eps, want_stats=True, out=None):
    """x_out = res + a @ w^T + bias (float32) and h = LayerNorm(x_out) * gamma + beta (bf16) in one kernel (star_gemm_res_ln).
    a [M, K], w [N, K] bf16; res float32 [M, N].  Returns (x_out, h, mean, rstd); `out` may be res."""
    _cuda(a, w, res, gamma, beta)
    m, k = a.shape
    n = w.shape[0]
    dt = L.dt_of(a)
    assert a.is_contiguous() and w.is_contiguous() and res.is_contiguous() and w.shape[1] == k and w.dtype == a.dtype
    assert res.shape == (m, n) and res.dtype == torch.float32 and bias.dtype == torch.float32 and bias.numel() == n
    assert gamma.dtype == torch.float32 and beta.dtype == torch.float32 and gamma.numel() == n and beta.numel() == n
    assert gamma.is_contiguous() and beta.is_contiguous() and bias.is_contiguous()
    if out is None:
        out = torch.empty_like(res)
    h = torch.empty(m, n, dtype=a.dtype, device=a.device)
    mean = torch.empty(m, dtype=torch.float32, device=a.device) if want_stats else None
    rstd = torch.empty(m, dtype=torch.float32, device=a.device) if want_stats else None
    with _timed("gemm_res_ln", 2.0 * m * n * k, float(m * k * 2 + n * k * 2 + m * n * (4 + 4 + 2))):
        L.check(L.lib().star_gemm_res_ln(p(a), p(w), p(bias), p(res), p(out), p(gamma), p(beta), float(eps), p(h), p(mean), p(rstd),
                                         m, n, k, dt, L.stream()))
    return out, h, mean, rstd


# ---------------------------------------------------------------------------------------------------
def attn_fwd(qkv, nb, l, h, hd, dt=None):
    """qkv [NB*L, 3*H*hd] -> (out [NB*L, H*hd], lse float32 [NB, H, L]).  dt = BF16X3: rows of bf16 hi/lo pairs, twice as wide."""
    _cuda(qkv)
    if dt is None:
        dt = L.dt_of(qkv)
    wd = L.width(dt)
    assert qkv.is_contiguous() and qkv.numel() == nb * l * 3 * h * hd * wd
    out = torch.empty(nb * l, h * hd * wd, dtype=qkv.dtype, device=qkv.device)
    lse = torch.empty(nb, h, l, dtype=torch.float32, device=qkv.device)
    with _timed("attn_fwd", 4.0 * nb * h * l * l * hd):
        L.check(L.lib().star_attn_fwd(p(qkv), p(out), p(lse), nb, l, h, hd, dt, L.stream()))
    return out, lse


def attn_bwd(qkv, out, dout, lse, nb, l, h, hd):
    """dqkv [NB*L, 3*H*hd] (star_attn_bwd: a dQ kernel and a dK/dV kernel, each recomputing the scores from the stored row
    log-sum-exp; no atomics, bitwise reproducible)."""
    dt = L.dt_of(qkv)
    assert qkv.is_contiguous() and out.is_contiguous() and dout.is_contiguous() and dout.dtype == qkv.dtype
    dqkv = torch.empty_like(qkv)
    delta = torch.empty(2, nb, h, l, dtype=torch.float32, device=qkv.device)   # workspace: delta | scaled lse
    with _timed("attn_bwd", 10.0 * nb * h * l * l * hd):
        L.check(L.lib().star_attn_bwd(p(qkv), p(out), p(dout), p(lse), p(delta), p(dqkv), nb, l, h, hd, dt, L.stream()))
    return dqkv


# ---------------------------------------------------------------------------------------------------
def patchify(img1, img_next, t, patch, dt, channels_last_order=False):
    """K1.  img1 float32 [BA,C,H,W] (any strides), img_next [BA,T-1,C,H,W] -> [BA*T*L, C*p*p]."""
    _cuda(img1)
    ba, c, h, w = img1.shape
    assert img1.dtype == torch.float32
    ln = (h // patch) * (w // patch)
    out = torch.empty(ba * t * ln, c * patch * patch, dtype=L.torch_dtype(dt), device=img1.device)
    sn = None
    if t > 1:
        assert img_next.dtype == torch.float32 and tuple(img_next.shape) == (ba, t - 1, c, h, w)
        sn = L.strides(img_next)
    with _timed("patchify", 0.0, float(ba) * t * c * h * w * (4 + out.element_size())):      # float32 voxels in, the patch matrix out
        L.check(L.lib().star_patchify(p(img1), L.strides(img1), p(img_next) if t > 1 else None, sn, p(out), ba, t, c, h, w,
                                      patch, dt | (256 if channels_last_order else 0), L.stream()))
    return out


def ce_argmax(logits, target, patch, dt, want_pred=False, want_dlogits=True, want_loss=True):
    """K9.  logits float32 [BA*L, 2*C*p*p] (patch layout), target float32 [BA,C,H,W] any strides.
    -> dict(loss float32[1], ind64, ind8, pred, dlogits)."""
    _cuda(logits, target)
    ba, c, h, w = target.shape
    assert logits.dtype == torch.float32 and logits.is_contiguous() and target.dtype == torch.float32
    dev = logits.device
    ind64 = torch.empty(ba, c, h, w, dtype=torch.int64, device=dev)
    ind8 = torch.empty(ba, c, h, w, dtype=torch.uint8, device=dev)
    pred = torch.empty(ba, 2, c, h, w, dtype=torch.float32, device=dev) if want_pred else None
    dlog = torch.empty(logits.shape, dtype=L.torch_dtype(dt), device=dev) if want_dlogits else None
    nb = L.lib().star_ce_nblocks(ba, c, h, w)
    part = torch.empty(nb, dtype=torch.float32, device=dev) if want_loss else None
    loss = torch.empty(1, dtype=torch.float32, device=dev) if want_loss else None
    # per voxel: two float32 logits + the float32 target read; the int64 decision (the API's ind_result) + its uint8 copy (the fusion's
    # input) written; + the two logit gradients in training, + the [BA,2,C,H,W] float32 logits tensor when a test asks for it.
    # (SURVEY 8d counts 16 B / voxel, a 4-byte decision word; PMC on MI355X: 409 MB fetched + 443 MB written per 40 frames = 25 B / voxel.)
    per_voxel = 8 + 4 + 8 + 1 + (2 * (2 if dt == BF16 else 4) if want_dlogits else 0) + (8 if want_pred else 0)
    with _timed("ce_argmax", 0.0, float(ba) * c * h * w * per_voxel):
        L.check(L.lib().star_ce_argmax(p(logits), p(target), L.strides(target), p(ind64), p(ind8), p(pred), p(dlog), p(part),
                                       ba, c, h, w, patch, dt, L.stream()))
    if want_loss:
        L.check(L.lib().star_ce_finish(p(part), nb, ba * c * h * w, p(loss), L.stream()))
    return dict(loss=loss, ind64=ind64, ind8=ind8, pred=pred, dlogits=dlog)


def fill_rows(x, fill, mask_token, temp0, pos, ba, l):
    d = x.shape[-1]
    L.check(L.lib().star_fill_rows(p(x), p(fill), p(mask_token), p(temp0), p(pos), ba, l, d, L.stream()))


def transpose_rows(src, pos, ba, l, d):
    dst = torch.empty_like(src)
    L.check(L.lib().star_transpose_rows(p(src), p(dst), p(pos), ba, l, d, L.stream()))
    return dst


def transpose_rows_bwd(gdst, ba, l, d):
    gsrc = torch.empty_like(gdst)
    L.check(L.lib().star_transpose_rows_bwd(p(gdst), p(gsrc), ba, l, d, L.stream()))
    return gsrc


def gather_rows(src, rows, dt=None, want_f32=True):
    """dst[m] = src[rows[m]] (zero row for -1); float32 and/or low-precision copies."""
    m, d = rows.numel(), src.shape[1]
    f32 = torch.empty(m, d, dtype=torch.float32, device=src.device) if want_f32 else None
    lp = torch.empty(m, d, dtype=L.torch_dtype(dt), device=src.device) if dt is not None else None
    L.check(L.lib().star_gather_rows(p(src), p(rows), p(f32), p(lp), m, d, dt if dt is not None else F32, L.stream()))
    return f32, lp


def scatter_rows(src, rows, out_rows):
    """dst[rows[m]] = src[m]; rows not hit stay zero.  src float32 or bf16 -> float32 [out_rows, D]."""
    m, d = src.shape
    dst = torch.zeros(out_rows, d, dtype=torch.float32, device=src.device)
    L.check(L.lib().star_scatter_rows(p(src), L.dt_of(src), p(rows), p(dst), m, d, L.stream()))
    return dst


def colsum_groups(src, div=1, mod=1, sel=None, sel_nonneg=False):
    """out[g] = sum of rows r of src with (r // div) % mod == g (and sel[r] selected) -> float32 [mod, D]."""
    r, d = src.shape
    assert src.is_contiguous()
    ws = torch.empty(64 * mod * d, dtype=torch.float32, device=src.device)
    out = torch.empty(mod, d, dtype=torch.float32, device=src.device)
    L.check(L.lib().star_colsum_groups(p(src), L.dt_of(src), p(sel), int(sel_nonneg), r, d, div, mod, p(ws), p(out), L.stream()))
    return out


def cast_transpose(src, dt):
    """float32 [R, C] -> dt [C, R]."""
    assert src.dtype == torch.float32 and src.is_contiguous() and src.dim() == 2
    r, c = src.shape
    dst = torch.empty(c, r, dtype=L.torch_dtype(dt), device=src.device)
    L.check(L.lib().star_cast_transpose(p(src), p(dst), r, c, dt, L.stream()))
    return dst


def cast(src, dt):
    assert src.dtype == torch.float32 and src.is_contiguous()
    dst = torch.empty(src.shape, dtype=L.torch_dtype(dt), device=src.device)
    L.check(L.lib().star_cast(p(src), p(dst), src.numel(), dt, L.stream()))
    return dst


def cast_batch_table(jobs, device):
    """jobs: list of (src float32 [R,C], dst, transposed[, extras]) with extras = dict(src_rows=int32 tensor [rows] (row gather: the
    job has that many rows), dst_ld, dst_off (elements), dst_f32).  -> (int64 device table, njobs, total_tiles) for cast_batch."""
    rows, start = [], 0
    for job in jobs:
        src, dst, tr = job[:3]
        ex = job[3] if len(job) > 3 else {}
        idx = ex.get("src_rows")
        r, c = (idx.numel() if idx is not None else src.shape[0]), src.shape[1]
        assert src.dtype == torch.float32 and src.is_contiguous() and dst.is_contiguous()
        assert idx is None or (idx.dtype == torch.int32 and idx.is_contiguous() and idx.device == dst.device)
        ld, off = int(ex.get("dst_ld", 0)), int(ex.get("dst_off", 0))
        last = off + ((c - 1) * (ld or r) + r if tr else (r - 1) * (ld or c) + c * (2 if ex.get("split") else 1))
        assert 0 <= off and last <= dst.numel(), "cast job writes past its destination"
        assert dst.dtype == (torch.float32 if ex.get("dst_f32") else dst.dtype) and (ex or dst.numel() == r * c)
        split = bool(ex.get("split"))      # bf16 hi/lo pairs (BF16X3): the lo values sit `c` elements further along each row
        assert not (split and (tr or ex.get("dst_f32"))) and (not split or ld >= 2 * c)
        rows.append([src.data_ptr(), dst.data_ptr(), r | (c << 32), int(bool(tr)) | (start << 32),      # StarCastJob, little endian
                     idx.data_ptr() if idx is not None else 0, ld | (off << 32), int(bool(ex.get("dst_f32"))) | (int(split) << 32)])
        start += ((r + 63) // 64) * ((c + 63) // 64)
    return torch.tensor(rows, dtype=torch.int64).to(device), len(jobs), start


def cast_batch(table, njobs, total_tiles, dt):
    L.check(L.lib().star_cast_batch(p(table), njobs, total_tiles, BF16 if dt == BF16X3 else dt, L.stream()))


def cast_split(src):
    """float32 [R, C] -> bf16 [R, 2C] hi/lo pairs (the BF16X3 storage of a weight)."""
    assert src.dtype == torch.float32 and src.is_contiguous() and src.dim() == 2
    r, c = src.shape
    dst = torch.empty(r, 2 * c, dtype=torch.bfloat16, device=src.device)
    tab, n, tiles = cast_batch_table([(src, dst, False, dict(dst_ld=2 * c, split=True))], src.device)
    cast_batch(tab, n, tiles, BF16)
    return dst


def cast_back(src):
    dst = torch.empty(src.shape, dtype=torch.float32, device=src.device)
    L.check(L.lib().star_cast_back(p(src), p(dst), src.numel(), L.dt_of(src), L.stream()))
    return dst


def adamw(pflat, gflat, m, v, wd_mask, lr, beta1, beta2, eps, wd, step, gscale=None):
    L.check(L.lib().star_adamw(p(pflat), p(gflat), p(m), p(v), p(wd_mask), p(gscale), pflat.numel(), float(lr), float(beta1),
                               float(beta2), float(eps), float(wd), int(step), L.stream()))


def sumsq(x):
    ws = torch.empty(1024, dtype=torch.float32, device=x.device)
    out = torch.empty(1, dtype=torch.float32, device=x.device)
    L.check(L.lib().star_sumsq(p(x), x.numel(), p(ws), p(out), L.stream()))
    return out


def iou_confusion(pred, target, conf=None):
    """K11.  pred/target float32 [N,Z,H,W] -> int64 [3,2,2] confusion counts (added into `conf`)."""
    _cuda(pred, target)
    pred, target = pred.contiguous(), target.contiguous()
    n, z, h, w = pred.shape
    if conf is None:
        conf = torch.zeros(3, 2, 2, dtype=torch.int64, device=pred.device)
    L.check(L.lib().star_iou_confusion(p(pred), p(target), p(conf), n, z, h, w, L.stream()))
    return conf


def voxel_scatter(idx, offsets, dims, channels_first=False, validate=True, out=None, host_offsets=None):
    """Sparse voxel indices -> dense frames.  idx int32/int64 [n,3] (x,y,z) and offsets int64 [F+1], both on the
    device; dims = (X, Y, Z).  -> float32 [F, Y, X, Z] (or [F, Z, Y, X]).  validate=True reads back the error word
    (one host sync) and raises IndexError like numpy's fancy assignment would.  host_offsets: the same offsets as a
    CPU int64 tensor (the loader has them anyway) -- lets large batches run in cache-sized chunks."""
    _cuda(idx, offsets)
    assert offsets.dtype == torch.int64 and offsets.dim() == 1 and offsets.is_contiguous()
    assert idx.dtype in (torch.int32, torch.int64) and idx.is_contiguous() and (idx.numel() == 0 or (idx.dim() == 2 and idx.shape[1] == 3))
    f = offsets.numel() - 1
    x, y, z = dims
    n = idx.shape[0] if idx.numel() else 0
    shape = (f, z, y, x) if channels_first else (f, y, x, z)
    if out is None:
        out = torch.empty(shape, dtype=torch.float32, device=offsets.device)
    assert out.dtype == torch.float32 and out.is_contiguous() and tuple(out.shape) == shape
    err = torch.zeros(1, dtype=torch.int32, device=offsets.device) if validate else None
    with _timed("voxel_scatter", 0.0, float(f) * x * y * z * 4 + float(n) * (3 * idx.element_size() + 4)):
        if host_offsets is not None:
            assert not host_offsets.is_cuda and host_offsets.dtype == torch.int64 and host_offsets.is_contiguous() and host_offsets.numel() == f + 1
        L.check(L.lib().star_voxel_scatter(p(idx) if n else None, int(idx.dtype == torch.int64), p(offsets), p(host_offsets), n, p(out), f,
                                           x, y, z, int(channels_first), p(err), L.stream()))
    if validate and int(err.item()) != 0:
        raise IndexError("voxel index out of bounds for dims %r" % (tuple(dims),))
    return out


# ---- VQ-STAR quantiser -----------------------------------------------------------------------------------
def vq_assign(x, codebook, dt, want_f32=True, want_lp=True, want_stats=True):
    """x float32 [M,E], codebook float32 [K,E] -> dict(idx int64 [M], q_f32, q_lp, part, counts)."""
    _cuda(x, codebook)
    assert x.dtype == torch.float32 and codebook.dtype == torch.float32 and x.is_contiguous() and codebook.is_contiguous()
    m, e = x.shape
    k = codebook.shape[0]
    assert codebook.shape[1] == e
    dev = x.device
    idx = torch.empty(m, dtype=torch.int64, device=dev)
    q32 = torch.empty(m, e, dtype=torch.float32, device=dev) if want_f32 else None
    qlp = torch.empty(m, e, dtype=L.torch_dtype(dt), device=dev) if want_lp else None
    nb = L.lib().star_vq_nblocks(m)
    part = torch.empty(nb, dtype=torch.float32, device=dev) if want_stats else None
    counts = torch.empty(k, dtype=torch.int32, device=dev) if want_stats else None
    with _timed("vq_assign", 2.0 * m * k * e, float(m * e * 4 + k * e * 4 + m * 8)):
        L.check(L.lib().star_vq_assign(p(x), p(codebook), p(idx), p(q32), p(qlp), p(part), p(counts), m, e, k, dt, L.stream()))
    return dict(idx=idx, q_f32=q32, q_lp=qlp, part=part, counts=counts)


def vq_finish(part, counts, m, e, commitment, loss_inout=None, ema=False):
    """-> float32 [3] = (vq loss, perplexity, mse); loss_inout[0] += vq loss.  ema: loss = commitment * mse alone."""
    out = torch.empty(3, dtype=torch.float32, device=part.device)
    L.check(L.lib().star_vq_finish(p(part), part.numel(), p(counts), m, e, counts.numel(), float(commitment), int(ema), p(out),
                                   p(loss_inout), L.stream()))
    return out


def vq_code_sums(x, idx, k):
    """-> float32 [K,E]: per-code sums of the rows of x assigned to it (deterministic order)."""
    m, e = x.shape
    dw = torch.empty(k, e, dtype=torch.float32, device=x.device)
    ws = torch.empty(L.lib().star_vq_bwd_ws(m, e, k), dtype=torch.float32, device=x.device)
    with _timed("vq_code_sums", 0.0, float(m * e * 4 + m * 8 + k * e * 4)):
        L.check(L.lib().star_vq_code_sums(p(x), p(idx), p(dw), p(ws), m, e, k, L.stream()))
    return dw


def vq_ema_update(counts, dw, codebook, ema_w, cluster_size, decay, epsilon=1e-5):
    """In-place EMA update of (cluster_size [K], ema_w [K,E], codebook [K,E]) from this batch's counts int32 [K] and dw [K,E]."""
    k, e = codebook.shape
    for t in (dw, codebook, ema_w, cluster_size):
        assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()
    assert counts.dtype == torch.int32 and counts.numel() == k and tuple(dw.shape) == (k, e) == tuple(ema_w.shape) and cluster_size.numel() == k
    L.check(L.lib().star_vq_ema_update(p(counts), p(dw), p(codebook), p(ema_w), p(cluster_size), e, k, float(decay), float(epsilon), L.stream()))


def vq_bwd(dq, x, idx, codebook, commitment, want_dlat=True, dcodebook=None, want_dcodebook=True):
    """-> (dlat in dq's dtype [M,E], dcodebook float32 [K,E])."""
    m, e = x.shape
    k = codebook.shape[0]
    dt = L.dt_of(dq)
    assert dq.is_contiguous() and tuple(dq.shape) == (m, e)
    dlat = torch.empty_like(dq) if want_dlat else None
    if dcodebook is None and want_dcodebook:
        dcodebook = torch.empty(k, e, dtype=torch.float32, device=x.device)
    assert dcodebook is None or (dcodebook.is_contiguous() and dcodebook.numel() == k * e and dcodebook.dtype == torch.float32)
    ws = torch.empty(L.lib().star_vq_bwd_ws(m, e, k), dtype=torch.float32, device=x.device) if dcodebook is not None else None
    with _timed("vq_bwd", 0.0, float(m * e * (4 + 2 * dq.element_size()) + 2 * k * e * 4)):
        L.check(L.lib().star_vq_bwd(p(dq), p(x), p(idx), p(codebook), p(dlat), p(dcodebook), p(ws), m, e, k, float(commitment), dt, L.stream()))
    return dlat, dcodebook


def onehot_rows(idx, k):
    out = torch.empty(idx.numel(), k, dtype=torch.float32, device=idx.device)
    L.check(L.lib().star_onehot_rows(p(idx), p(out), idx.numel(), k, L.stream()))
    return out


# ---- joint (latent-level fusion) variant: feature warp, conv-head pieces ---------------------------------------------
def feat_warp_fuse(feat, theta, num_agent, batch, grid, pos=None):
    """feat float32 [A*B, G*G, D] -> (fused float32 [A*B, G*G, D] (+ pos), passthrough uint8 [A*B])."""
    _cuda(feat, theta, num_agent)
    ab, l, d = feat.shape
    a = ab // batch
    assert feat.dtype == torch.float32 and feat.is_contiguous() and l == grid * grid
    assert theta.dtype == torch.float32 and theta.is_contiguous() and tuple(theta.shape) == (batch, a, a, 8)
    assert num_agent.dtype in (torch.int32, torch.int64) and num_agent.is_contiguous() and tuple(num_agent.shape) == (batch, a)
    assert pos is None or (pos.dtype == torch.float32 and pos.is_contiguous() and tuple(pos.shape) == (l, d))
    out = torch.empty_like(feat)
    pt = torch.empty(ab, dtype=torch.uint8, device=feat.device)
    L.check(L.lib().star_feat_warp_fuse(p(feat), p(theta), p(num_agent), int(num_agent.dtype == torch.int64), p(pos), p(out), p(pt), batch, a,
                                        grid, d, L.stream()))
    return out, pt


def upsample2x(x, n, h, w):
    """channels-last [n*h*w, C] (bf16 / float32) -> [n*2h*2w, C], bilinear, align_corners=False."""
    c = x.shape[-1]
    assert x.is_contiguous() and x.numel() == n * h * w * c
    y = torch.empty(n * 4 * h * w, c, dtype=x.dtype, device=x.device)
    L.check(L.lib().star_upsample2x_fwd(p(x), p(y), n, h, w, c, L.dt_of(x), L.stream()))
    return y


def upsample2x_bwd(dy, n, h, w):
    """float32 gradient [n*2h*2w, C] -> [n*h*w, C]."""
    c = dy.shape[-1]
    assert dy.dtype == torch.float32 and dy.is_contiguous() and dy.numel() == n * 4 * h * w * c
    dx = torch.empty(n * h * w, c, dtype=torch.float32, device=dy.device)
    L.check(L.lib().star_upsample2x_bwd(p(dy), p(dx), n, h, w, c, L.stream()))
    return dx


def bn_stats(x, eps):
    """x float32 [R, ld] -> (mean, biased var, rstd) float32 [ld]."""
    r, ld = x.shape
    assert x.dtype == torch.float32 and x.is_contiguous()
    mean, var, rstd = (torch.empty(ld, dtype=torch.float32, device=x.device) for _ in range(3))
    ws = torch.empty(L.lib().star_bn_ws(ld), dtype=torch.float32, device=x.device)
    L.check(L.lib().star_bn_stats(p(x), r, ld, float(eps), p(mean), p(var), p(rstd), p(ws), L.stream()))
    return mean, var, rstd


def bn_apply(x, mean, rstd, gamma, beta, relu, dt):
    r, ld = x.shape
    y = torch.empty(r, ld, dtype=L.torch_dtype(dt), device=x.device)
    L.check(L.lib().star_bn_apply(p(x), p(mean), p(rstd), p(gamma), p(beta), int(relu), r, ld, p(y), dt, L.stream()))
    return y


def bn_bwd(x, dy, mean, rstd, gamma, beta, relu, training, dt):
    """-> (dx in dt [R, ld], dgamma, dbeta float32 [ld])."""
    r, ld = x.shape
    assert dy.dtype == torch.float32 and dy.is_contiguous() and tuple(dy.shape) == (r, ld)
    dx = torch.empty(r, ld, dtype=L.torch_dtype(dt), device=x.device)
    dgamma = torch.empty(ld, dtype=torch.float32, device=x.device)
    dbeta = torch.empty(ld, dtype=torch.float32, device=x.device)
    ws = torch.empty(L.lib().star_bn_ws(ld), dtype=torch.float32, device=x.device)
    L.check(L.lib().star_bn_bwd(p(x), p(dy), p(mean), p(rstd), p(gamma), p(beta), int(relu), int(training), r, ld, p(dx), dt, p(dgamma), p(dbeta),
                                p(ws), L.stream()))
    return dx, dgamma, dbeta


def ce_nhwc(lfree, locc, target, c, want_pred=True, want_grad=True):
    """Two heads' channels-last logits float32 [N*H*W, ld], target float32 [N,C,H,W] (any strides)
    -> dict(loss [1], ind64 [N,C,H,W], pred [N,2,C,H,W], dfree, docc)."""
    n, ct, h, w = target.shape
    ld = lfree.shape[1]
    assert ct == c and lfree.dtype == torch.float32 and lfree.is_contiguous() and locc.is_contiguous() and tuple(lfree.shape) == (n * h * w, ld) == tuple(locc.shape)
    dev = lfree.device
    ind = torch.empty(n, c, h, w, dtype=torch.int64, device=dev)
    pred = torch.empty(n, 2, c, h, w, dtype=torch.float32, device=dev) if want_pred else None
    dfree = torch.empty_like(lfree) if want_grad else None
    docc = torch.empty_like(locc) if want_grad else None
    nb = L.lib().star_ce_nhwc_blocks(n, h, w)
    part = torch.empty(nb, dtype=torch.float32, device=dev)
    loss = torch.empty(1, dtype=torch.float32, device=dev)
    L.check(L.lib().star_ce_nhwc(p(lfree), p(locc), p(target), L.strides(target), p(ind), p(pred), p(dfree), p(docc), p(part), n, c, h, w, ld, L.stream()))
    L.check(L.lib().star_ce_finish(p(part), nb, n * c * h * w, p(loss), L.stream()))
    return dict(loss=loss, ind64=ind, pred=pred, dfree=dfree, docc=docc)
