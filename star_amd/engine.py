"""Forward / backward orchestration of the STAR amortized-individual model over libstar_hip.so.

This is the host side of the hot path (reference: AmortizedIndivMMAEViT.forward / .inference,
star/models/multiagent_mae.py:946-1125).  Every arithmetic step is one C-ABI call (star_amd.ops); PyTorch
allocates buffers and keeps the autograd graph edge (one autograd.Function for the whole model), nothing else.

Row conventions (all matrices are [rows, features], rows contiguous):
  frame f = ba*T + t      (agent-scene ba, timestamp t)       -- multiagent_mae.py:960
  token row = f*L + tok   encoder side,  ba*L + tok decoder side
  compressed row r = f*keep + j
"""
import math
import os

import torch

from . import ops
from ._lib import ACT_DRELU, ACT_GELU, ACT_MULAUX, ACT_NONE, ACT_RELU, BF16, BF16X3, F32, GEMM_A_PLAIN, GEMM_PRE_IS_GRAD, GEMM_RES_BY_AROW

_LINEARS_BLOCK = ("attn.qkv", "attn.proj", "mlp.fc1", "mlp.fc2")
SIDE_STREAM = os.environ.get("STAR_AMD_SIDE_STREAM", "1") != "0"   # weight gradients on a second stream


class Config:
    def __init__(self, model):
        pe = model.patch_embed
        self.img = pe.img_size[0]
        self.patch = pe.patch_size[0]
        self.chans = model.in_chans
        self.dim = model.pos_embed.shape[-1]
        self.depth = len(model.blocks)
        self.heads = model.blocks[0].attn.num_heads
        self.ddim = model.decoder_pos_embed.shape[-1]
        self.ddepth = len(model.decoder_blocks)
        self.dheads = model.decoder_blocks[0].attn.num_heads
        self.T = model.time_stamp
        self.mask = model.mask_method
        self.encode_partial = model.encode_partial
        self.no_temp_emb = model.no_temp_emb
        self.decode_singletemp = model.decode_singletemp
        self.eps = model.blocks[0].norm1.eps
        self.inter = model.compressor[0].out_features
        vq = getattr(model, "_vq_star", None)                 # VQ-STAR variant (star/models/VQSTAR.py)
        self.vq = vq is not None
        self.commitment = float(vq._commitment_cost) if vq is not None else 0.0
        self.vq_decay = float(getattr(vq, "_decay", 0.0)) if vq is not None else 0.0    # > 0: EMA codebook (mae_base.py:176)
        self.vq_epsilon = float(getattr(vq, "_epsilon", 1e-5)) if vq is not None else 0.0
        self.grid = self.img // self.patch
        self.L = self.grid * self.grid
        self.V = self.chans * self.patch * self.patch


_HEAD_PERM = {}


def head_permutation(cfg, device, as_int64=False):
    """Row order of the fused prediction-head weight.  Reference rows are (py, px, c) per class
    (mae_base.py:392-406 un-patchify order); the kernels want class-major (c, py, px) so that the 8
    voxels of a patch row are contiguous.  Returns idx with new_row[n] = old_row[idx[n]] (int32 for the cast jobs, int64 for torch
    indexing)."""
    p, c = cfg.patch, cfg.chans
    key = (p, c, str(device))
    key = key + (bool(as_int64),)
    idx = _HEAD_PERM.get(key)
    if idx is None:   # built once: a host-to-device copy in the middle of the backward pass would stall the launch queue
        cc, py, px = torch.meshgrid(torch.arange(c), torch.arange(p), torch.arange(p), indexing="ij")
        idx = _HEAD_PERM[key] = ((py * p + px) * c + cc).reshape(-1).to(torch.int64 if as_int64 else torch.int32).to(device)
    return idx


class Weights:
    """Low-precision copies of the parameters in the layouts the kernels read, refreshed when a
    parameter's version counter changes (every optimizer step in training, never in inference) or when
    `epoch` is bumped (the fused optimizer updates the flat parameter buffer behind autograd's back)."""

    def __init__(self):
        self.cache = {}
        self.epoch = 0
        self.plain = {}      # key -> entry: copies that are cast jobs, refreshed together in ONE launch
        self.table = None    # (device job table, njobs, tiles, dt, signature)

    def jobs_entry(self, key, params, dt, make):
        """The copy registered under `key` (key[1] = dt), built from `params` by cast jobs.  make() -> (dst, jobs) allocates the
        destination and lists its jobs (ops.cast_batch_table tuples) on first use.  A stale copy triggers ONE batched launch that
        refreshes every registered copy (they all go stale together: one optimizer step)."""
        versions = tuple(q._version for q in params) + (self.epoch,)
        ptrs = tuple(q.data_ptr() for q in params)
        ent = self.plain.get(key)
        if ent is not None and ent["ptrs"] == ptrs and ent["versions"] == versions:
            return ent["dst"]
        if ent is None or ent["ptrs"] != ptrs or ent["dst"].device != params[0].device:
            dst, jobs = make()
            tab, n, tiles = ops.cast_batch_table(jobs, params[0].device)
            ops.cast_batch(tab, n, tiles, dt)
            self.plain[key] = dict(ptrs=ptrs, params=params, dst=dst, jobs=jobs, versions=versions)
            self.table = None
            return dst
        same_dt = [e for k, e in self.plain.items() if k[1] == dt]
        sig = tuple((e["ptrs"], e["dst"].data_ptr()) for e in same_dt)
        if self.table is None or self.table[3] != dt or self.table[4] != sig:
            tab, n, tiles = ops.cast_batch_table([j for e in same_dt for j in e["jobs"]], params[0].device)
            self.table = (tab, n, tiles, dt, sig)
        ops.cast_batch(self.table[0], self.table[1], self.table[2], dt)
        for e in same_dt:
            e["versions"] = tuple(q._version for q in e["params"]) + (self.epoch,)
        return ent["dst"]

    def linear(self, key, w, dt, transposed):
        """The dt copy of weight `w` ([N,K], or [K,N] when transposed)."""
        def make():
            n, k = w.shape
            if dt == BF16X3:       # rows of [hi | lo] (forward only: no transposed copies)
                assert not transposed
                dst = torch.empty(n, 2 * k, dtype=torch.bfloat16, device=w.device)
                return dst, [(w.detach(), dst, False, dict(dst_ld=2 * k, split=True))]
            dst = torch.empty((k, n) if transposed else (n, k), dtype=ops.L.torch_dtype(dt), device=w.device)
            return dst, [(w.detach(), dst, transposed)]
        return self.jobs_entry(key, (w,), dt, make)

    def get(self, key, versions, build):
        versions = (versions, self.epoch)
        hit = self.cache.get(key)
        if hit is not None and hit[0] == versions:
            return hit[1]
        val = build()
        self.cache[key] = (versions, val)
        return val


def _lin(P, W, name, dt, transposed=False):
    w = P[name + ".weight"]
    return W.linear((name, dt, transposed), w, dt, transposed)


def _heads(P, W, cfg, dt, transposed=False):
    """The two prediction heads as ONE Linear in the kernels' class-major row order: rows [0, V) = decoder_pred_free, [V, 2V) =
    decoder_pred_occ, each gathered through head_permutation -- two jobs of the step's batched weight refresh (no launches of
    their own)."""
    wo, wf = P["decoder_pred_occ.weight"], P["decoder_pred_free.weight"]

    def make():
        idx = head_permutation(cfg, wo.device)
        v, k = wf.shape
        if dt == BF16X3:
            assert not transposed
            dst = torch.empty(2 * v, 2 * k, dtype=torch.bfloat16, device=wo.device)
            return dst, [(wf.detach(), dst, False, dict(src_rows=idx, dst_ld=2 * k, dst_off=0, split=True)),
                         (wo.detach(), dst, False, dict(src_rows=idx, dst_ld=2 * k, dst_off=v * 2 * k, split=True))]
        dst = torch.empty((k, 2 * v) if transposed else (2 * v, k), dtype=ops.L.torch_dtype(dt), device=wo.device)
        if transposed:
            return dst, [(wf.detach(), dst, True, dict(src_rows=idx, dst_ld=2 * v, dst_off=0)),
                         (wo.detach(), dst, True, dict(src_rows=idx, dst_ld=2 * v, dst_off=v))]
        return dst, [(wf.detach(), dst, False, dict(src_rows=idx, dst_ld=k, dst_off=0)),
                     (wo.detach(), dst, False, dict(src_rows=idx, dst_ld=k, dst_off=v * k))]

    return W.jobs_entry(("heads", dt, transposed), (wf, wo), dt, make)


def _heads_bias(P, W, cfg, dt):
    """float32 bias of the fused heads, same row order; rides in the batched refresh of the dt copies."""
    bo, bf = P["decoder_pred_occ.bias"], P["decoder_pred_free.bias"]

    def make():
        idx = head_permutation(cfg, bo.device)
        v = bf.numel()
        dst = torch.empty(2 * v, dtype=torch.float32, device=bo.device)
        return dst, [(bf.detach().reshape(v, 1), dst, False, dict(src_rows=idx, dst_ld=1, dst_off=0, dst_f32=True)),
                     (bo.detach().reshape(v, 1), dst, False, dict(src_rows=idx, dst_ld=1, dst_off=v, dst_f32=True))]

    return W.jobs_entry(("heads_bias", dt, False), (bf, bo), dt, make)


def _patch_weight(P, W, cfg, dt, channels_last):
    w = P["patch_embed.proj.weight"]

    def build():
        w2 = w.detach()
        w2 = w2.permute(0, 2, 3, 1) if channels_last else w2     # (py, px, c) order for NHWC memory
        w2 = w2.reshape(w.shape[0], -1).contiguous()
        return ops.cast_split(w2) if dt == BF16X3 else ops.cast(w2, dt)

    return W.get(("patch", dt, channels_last), w._version, build)


# -------------------------------------------------------------------------------------------------------
def _block_fwd(x, P, W, pre, heads, nb, l, dt, eps, save, train, ln1=None, nxt=None):
    """One transformer block.  ln1: (h1, mean1, rstd1) when the kernel that produced x already normalised it; nxt: name of the
    LayerNorm applied to this block's result (the next block's norm1 or the final norm) -- when given, returns (x_out, ln) with
    ln = (h, mean, rstd) of that LayerNorm (from the fc2 kernel where star_gemm_res_ln covers the shape), else x_out alone."""
    d = x.shape[1]
    hd = d // heads
    fuse_ln = _GEMM_LN and dt == BF16 and ops.gemm_res_ln_prefers(x.shape[0], d, d, dt, x, P[pre + "attn.proj.bias"], P[pre + "norm2.weight"],
                                                                  P[pre + "norm2.bias"])
    h1, mean1, rstd1 = ln1 if ln1 is not None else ops.layernorm_fwd(x, P[pre + "norm1.weight"], P[pre + "norm1.bias"], eps, dt, want_stats=train)
    qkv, _, _ = ops.gemm(h1, _lin(P, W, pre + "attn.qkv", dt), bias=P[pre + "attn.qkv.bias"], dt=dt)
    a, lse = ops.attn_fwd(qkv, nb, l, heads, hd, dt=dt)
    if fuse_ln:      # x_mid = x + proj(a) and norm2(x_mid) in one launch
        x_mid, h2, mean2, rstd2 = ops.gemm_res_ln(a, _lin(P, W, pre + "attn.proj", dt), P[pre + "attn.proj.bias"], x, P[pre + "norm2.weight"],
                                                  P[pre + "norm2.bias"], eps, want_stats=train, out=None if train else x)
    else:
        _, x_mid, _ = ops.gemm(a, _lin(P, W, pre + "attn.proj", dt), bias=P[pre + "attn.proj.bias"], res=x, want_lp=False,
                               want_f32=True, out_f32=None if train else x, dt=dt)
        h2, mean2, rstd2 = ops.layernorm_fwd(x_mid, P[pre + "norm2.weight"], P[pre + "norm2.bias"], eps, dt, want_stats=train)
    # u is not the pre-activation but gelu'(pre-activation): all the backward needs of it (flag PRE_IS_GRAD)
    w1, w2 = _lin(P, W, pre + "mlp.fc1", dt), _lin(P, W, pre + "mlp.fc2", dt)
    ln = None
    v, _, u = ops.gemm(h2, w1, bias=P[pre + "mlp.fc1.bias"], act=ACT_GELU, want_pre=train, flags=GEMM_PRE_IS_GRAD, dt=dt)
    if nxt is not None and _GEMM_LN and dt == BF16 and ops.gemm_res_ln_prefers(x.shape[0], d, w1.shape[0], dt, P[pre + "mlp.fc2.bias"],
                                                                               P[nxt + ".weight"], P[nxt + ".bias"]):
        x_out, hn, mean_n, rstd_n = ops.gemm_res_ln(v, w2, P[pre + "mlp.fc2.bias"], x_mid, P[nxt + ".weight"], P[nxt + ".bias"], eps,
                                                    want_stats=train, out=None if train else x_mid)
        ln = (hn, mean_n, rstd_n)
    else:
        _, x_out, _ = ops.gemm(v, w2, bias=P[pre + "mlp.fc2.bias"], res=x_mid, want_lp=False, want_f32=True,
                               out_f32=None if train else x_mid, dt=dt)
    if train:
        save.append(dict(pre=pre, x=x, mean1=mean1, rstd1=rstd1, h1=h1, qkv=qkv, a=a, lse=lse, x_mid=x_mid, mean2=mean2,
                         rstd2=rstd2, h2=h2, u=u, v=v, heads=heads, nb=nb, l=l))
    if nxt is None:
        return x_out
    if ln is None:
        ln = ops.layernorm_fwd(x_out, P[nxt + ".weight"], P[nxt + ".bias"], eps, dt, want_stats=train)
    return x_out, ln


def _blocks_fwd(x, P, W, fmt, depth, norm, heads, nb, l, dt, eps, save, train, taps=None, tap=None):
    """`depth` blocks named fmt % i and the LayerNorm `norm` behind them -> (x, y, mean, rstd).  Every LayerNorm but the first is
    produced by the Linear in front of it (star_gemm_res_ln) where that kernel covers the shape."""
    ln = None
    for i in range(depth):
        nxt = (fmt % (i + 1)) + "norm1" if i + 1 < depth else norm
        x, ln = _block_fwd(x, P, W, fmt % i, heads, nb, l, dt, eps, save, train, ln1=ln, nxt=nxt)
        if taps is not None:
            taps[tap % i] = x.clone()
    if ln is None:      # no blocks at all
        ln = ops.layernorm_fwd(x, P[norm + ".weight"], P[norm + ".bias"], eps, dt, want_stats=train)
    return (x,) + tuple(ln)


def _put(grads, G, name, val):
    """record a gradient that no kernel wrote in place (copied into the caller's view when given)."""
    if G is not None:
        G[name].copy_(val.reshape(G[name].shape))
        val = G[name]
    grads[name] = val


def stages(cfg):
    """Parameter names grouped in the order their gradients complete in backward(); only parameters the
    configured path really uses (the others never receive a gradient, reference Q5).  Every LayerNorm weight / bias sits in the
    LAST stage: their gradients are folded by one launch at the end of the backward (_LnFolds)."""
    blk = ["attn.qkv.weight", "attn.qkv.bias", "attn.proj.weight", "attn.proj.bias",
           "mlp.fc1.weight", "mlp.fc1.bias", "mlp.fc2.weight", "mlp.fc2.bias"]
    lnb = ["norm1.weight", "norm1.bias", "norm2.weight", "norm2.bias"]
    norms = ["decoder_norm.weight", "decoder_norm.bias"]
    out = [["decoder_pred_occ.weight", "decoder_pred_occ.bias", "decoder_pred_free.weight", "decoder_pred_free.bias"]]
    for i in reversed(range(cfg.ddepth)):
        out.append(["decoder_blocks.%d.%s" % (i, n) for n in blk])
        norms += ["decoder_blocks.%d.%s" % (i, n) for n in lnb]
    mid = []
    if cfg.mask == "random" or cfg.decode_singletemp:
        mid.append("mask_token")
    if not cfg.no_temp_emb:
        mid.append("decoder_temp_embed")
    mid += ["decoder_embed.weight", "decoder_embed.bias", "decompressor.weight", "decompressor.bias"]
    if getattr(cfg, "vq", False) and not getattr(cfg, "vq_decay", 0.0) > 0.0:     # an EMA codebook is not a trained parameter
        mid.append("_vq_star._embedding.weight")
    mid += ["compressor.0.weight", "compressor.0.bias"]
    norms += ["norm.weight", "norm.bias"]
    out.append(mid)
    for i in reversed(range(cfg.depth)):
        out.append(["blocks.%d.%s" % (i, n) for n in blk])
        norms += ["blocks.%d.%s" % (i, n) for n in lnb]
    out.append(["patch_embed.proj.weight", "patch_embed.proj.bias"] + norms)
    return out


class _SideStream:
    """Weight-gradient GEMMs do not feed the rest of the backward chain, so they run on a second HIP stream and
    share the CUs with the latency-bound kernels of the main chain (attention backward, LayerNorm).  `join` makes
    the caller's stream wait for everything issued here (before a gradient bucket is reduced / the step ends)."""
    streams = {}

    def __init__(self, device):
        self.device = device
        if device not in _SideStream.streams:
            prio = os.environ.get("STAR_AMD_SIDE_PRIO")
            _SideStream.streams[device] = torch.cuda.Stream(device=device) if prio is None else torch.cuda.Stream(device=device, priority=int(prio))
        self.stream = _SideStream.streams[device]
        self.used = False

    def run(self, fn, *tensors, wait=True):
        """wait=False: every input was produced BEFORE an earlier run() on this stream waited for the caller's stream (a cross-stream
        wait costs the caller's stream ~10 us: tools/probe_event_cost.py)."""
        main = torch.cuda.current_stream(self.device)
        if wait:
            self.stream.wait_stream(main)             # inputs are produced on the caller's stream
        with torch.cuda.stream(self.stream):
            out = fn()
        for t in tensors:
            if t is not None:
                t.record_stream(self.stream)          # keep the caching allocator from recycling them early
        self.used = True
        return out

    def join(self):
        if self.used:
            torch.cuda.current_stream(self.device).wait_stream(self.stream)
            self.used = False


_GEMM_LN = os.environ.get("STAR_AMD_GEMM_LN", "1") == "1"           # proj / fc2 + residual + the LayerNorm behind them in one launch (star_gemm_res_ln)
_SKIP_WGRAD = os.environ.get("STAR_AMD_SKIP_WGRAD", "0") == "1"   # measurement only (wrong gradients): the backward chain without its weight-gradient GEMMs


def _wg(grads, G, name, dy, x, x_rows=None, side=None, wait=True, queue=None):
    """weight + bias gradient of Linear `name`, written into the caller's gradient views when given.  queue (a _WgQueue): the
    product waits there for the next grouped launch (star_gemm_wgrad_group) when it is eligible for one."""
    if _SKIP_WGRAD and G is not None:
        grads[name + ".weight"], grads[name + ".bias"] = G[name + ".weight"], G[name + ".bias"]
        return
    if queue is not None and ops.wgrad_group_eligible(dy, x):
        # a row gather (the compressor reads the kept tokens only) becomes a gathered COPY of x, made on the launching stream right
        # before the grouped launch: pure data movement (M x K bf16), which keeps the product in the grouped kernel; rows with
        # x_rows < 0 are padding slots whose dy rows are zero (their outputs are dropped on the way to the decoder), so any row serves
        xg = x if x_rows is None else (lambda: x.index_select(0, x_rows.clamp(min=0).long()))
        grads[name + ".weight"], grads[name + ".bias"] = queue.add(dy, xg, None if G is None else G[name + ".weight"],
                                                                   None if G is None else G[name + ".bias"], keep=(x, x_rows))
        return

    def run():
        return ops.gemm_wgrad(dy, x, x_rows=x_rows, dw=None if G is None else G[name + ".weight"],
                              db=None if G is None else G[name + ".bias"])
    dw, db = run() if side is None else side.run(run, dy, x, x_rows, wait=wait)
    grads[name + ".weight"], grads[name + ".bias"] = dw, db


# Weight gradients of a backward pass leave in GROUPS: one launch computes up to eight of them and folds the split partials in the
# kernel (ops.gemm_wgrad_group; tile shape and split count depend on each product alone, so the grouping never changes a bit of the
# result).  Where the groups are cut (STAR_AMD_WGRAD_FLUSH, a comma list of points inside a block's backward):
#   "block" (default) at the end of every block: its four products, one launch beside the next block's input-gradient GEMMs;
#   "attn"  right before every attention backward (a block's fc2 / fc1 / proj products and the qkv product of the block before it);
#   "fc1", "proj": after that product is queued (finer groups).
# Measured inside one GPU call (B = 8 training step, round 3, final kernel: 384 x 192 / 256 x 256 tiles): "block" 11.68-11.70 ms, "attn"
# 11.67, "proj" 11.61 -- the step does not follow the flush point, the tile or the split count: what the
# side stream overlaps stretches by what it takes.  The round-2 form (separate launches + reduce kernels, 90 launches) measured 11.52-11.57.
# A data-parallel wrapper (on_ready) needs a stage's gradients complete at the stage's end: the queue is flushed there too.
_DP_JOIN = os.environ.get("STAR_AMD_DP_JOIN", "0") == "1"
WG_GROUP = os.environ.get("STAR_AMD_WGRAD_GROUP", "1") == "1"
WG_FLUSH = os.environ.get("STAR_AMD_WGRAD_FLUSH", "block")
_WG_TAIL_MAIN = os.environ.get("STAR_AMD_WGRAD_TAIL_MAIN", "1") == "1"
_WG_FLUSH_AT = set(WG_FLUSH.split(","))       # any of: fc1 (after the fc1 product is queued), proj, attn, block


class _WgQueue:
    def __init__(self, side):
        self.side, self.jobs, self.post = side, [], []

    def add(self, dy, x, dw, db, post=None, keep=()):
        """x: the operand, or a callable that makes it on the launching stream; keep: tensors the callable reads (kept alive, and
        recorded on the side stream)"""
        dev = dy.device
        k = keep[0].shape[1] if callable(x) else x.shape[1]
        if dw is None:
            dw = torch.empty(dy.shape[1], k, dtype=torch.float32, device=dev)
        if db is None:
            db = torch.empty(dy.shape[1], dtype=torch.float32, device=dev)
        self.jobs.append((dy, x, dw, db, keep))
        if post is not None:
            self.post.append(post)
        return dw, db

    def flush(self, at=None):
        if not self.jobs or (at is not None and at not in _WG_FLUSH_AT):
            return
        jobs, post = self.jobs, self.post
        self.jobs, self.post = [], []

        def run():
            ops.gemm_wgrad_group([(dy, x() if callable(x) else x, dw, db) for dy, x, dw, db, _ in jobs])
            for fn in post:
                fn()
        if self.side is None:
            run()
        else:
            self.side.run(run, *[t for j in jobs for t in (j[0], j[1], *j[4]) if isinstance(t, torch.Tensor)])


def _ln_bwd(grads, G, name, dy, x, P, mean, rstd, g_in=None, fold=None):
    """LayerNorm backward.  `fold` (an _LnFolds): the dgamma / dbeta fold is deferred to ONE launch at the end of the backward."""
    kw = dict(g_in=g_in, dgamma=None if G is None else G[name + ".weight"], dbeta=None if G is None else G[name + ".bias"])
    if fold is not None and G is not None:
        kw["defer_ws"] = fold.workspace(name, x)
    g, g_lp, dgam, dbet = ops.layernorm_bwd(dy, x, P[name + ".weight"], mean, rstd, **kw)
    if fold is not None and G is not None:
        fold.add(name, dgam, dbet, x)
    grads[name + ".weight"], grads[name + ".bias"] = dgam, dbet
    return g, g_lp


class _LnFolds:
    """The deferred dgamma / dbeta folds of one backward pass: a persistent workspace per LayerNorm (kept on the model's Weights
    object: 2.4 MB each at D = 384) and a device job table that is rebuilt only when a pointer or a row count changes."""

    def __init__(self, W):
        self.W = W
        if not hasattr(W, "ln_ws"):
            W.ln_ws, W.ln_table = {}, None
        self.jobs = []

    def workspace(self, name, x):
        need = ops.L.lib().star_layernorm_bwd_ws(x.shape[1])
        ws = self.W.ln_ws.get(name)
        if ws is None or ws.numel() < need or ws.device != x.device:
            ws = self.W.ln_ws[name] = torch.empty(need, dtype=torch.float32, device=x.device)
        return ws

    def add(self, name, dgam, dbet, x):
        self.jobs.append((self.W.ln_ws[name], dgam, dbet, x.shape[0], x.shape[1]))

    def run(self):
        if not self.jobs:
            return
        sig = tuple((j[0].data_ptr(), j[1].data_ptr(), j[2].data_ptr(), j[3], j[4]) for j in self.jobs)
        if self.W.ln_table is None or self.W.ln_table[0] != sig:
            self.W.ln_table = (sig,) + ops.ln_fold_table(self.jobs, self.jobs[0][0].device)
        _, tab, n, max_d = self.W.ln_table
        ops.ln_fold_batch(tab, n, max_d)
        self.jobs = []


# The fc2 weight gradient of a block is queued on the side stream right before the block's attention backward (compute-bound, the
# best partner for it) instead of beside the memory-bound dgrad GEMMs and LayerNorm in between; the other three start as soon as
# their operands exist.  A/B inside one GPU call: this order 12.75 ms/step, all three early ones deferred 12.85, none 12.92; on the final
# round-2 build: this order 11.72, fc1 instead of fc2 held back 11.85, both 11.85.
WG_DEFER = os.environ.get("STAR_AMD_WG_DEFER", "1") == "1"
# the weight gradients outside the blocks (prediction heads, decoder_embed, decompressor, compressor) run on the side stream too:
# A/B inside one GPU call 13.51 -> 13.35 ms/step
SIDE_EXTRA = os.environ.get("STAR_AMD_SIDE_EXTRA", "1") == "1"
LN_FOLD_BATCH = os.environ.get("STAR_AMD_LN_FOLD_BATCH", "1") == "1"   # 0: every LayerNorm folds its dgamma / dbeta rows at once (A/B)


def _block_bwd(g, g_lp, s, P, W, dt, grads, G=None, side=None, fold=None, queue=None):
    pre = s["pre"]
    heads, nb, l = s["heads"], s["nb"], s["l"]
    hd = s["x"].shape[1] // heads
    late = []
    wg = (lambda *a, **k: late.append((a, k))) if (WG_DEFER and side is not None and queue is None) else _wg
    wg(grads, G, pre + "mlp.fc2", g_lp, s["v"], side=side, queue=queue)
    du, _, _ = ops.gemm(g_lp, _lin(P, W, pre + "mlp.fc2", dt, True), act=ACT_MULAUX, aux=s["u"])
    _wg(grads, G, pre + "mlp.fc1", du, s["h2"], side=side, queue=queue)
    if queue is not None:
        queue.flush("fc1")
    dh2, _, _ = ops.gemm(du, _lin(P, W, pre + "mlp.fc1", dt, True))
    g1, g1_lp = _ln_bwd(grads, G, pre + "norm2", dh2, s["x_mid"], P, s["mean2"], s["rstd2"], g_in=g, fold=fold)
    _wg(grads, G, pre + "attn.proj", g1_lp, s["a"], side=side, queue=queue)
    if queue is not None:
        queue.flush("proj")
    da, _, _ = ops.gemm(g1_lp, _lin(P, W, pre + "attn.proj", dt, True))
    for a, k in late:          # the held-back fc2 weight gradient starts beside the attention backward; its operands are older than
        _wg(*a, wait=False, **k)   # the side stream's last wait (fc1 / proj above), so it needs no cross-stream wait of its own
    if queue is not None:
        queue.flush("attn")
    dqkv = ops.attn_bwd(s["qkv"], s["a"], da, s["lse"], nb, l, heads, hd)
    _wg(grads, G, pre + "attn.qkv", dqkv, s["h1"], side=side, queue=queue)
    if queue is not None:
        queue.flush("block")
    dh1, _, _ = ops.gemm(dqkv, _lin(P, W, pre + "attn.qkv", dt, True))
    g0, g0_lp = _ln_bwd(grads, G, pre + "norm1", dh1, s["x"], P, s["mean1"], s["rstd1"], g_in=g1, fold=fold)
    return g0, g0_lp


# -------------------------------------------------------------------------------------------------------
def forward(cfg, P, W, dt, imgs1, imgs_next, trans_matrices, num_agent, batch, mask_ratio, infer, train, noise=None, taps=None,
            vq_update=False, theta=None):
    """Runs the model.  Returns (out, ctx): out = dict(loss [1], result, latent, ind); ctx feeds backward()."""
    dev = imgs1.device
    if dt == BF16X3 and (train or cfg.vq):
        raise RuntimeError("star_amd: compute dtype 'bf16x3' (bf16 hi/lo pairs) is an inference mode of the STAR completion model: "
                           "train in 'bf16' or 'fp32', and run VQ-STAR in those")
    ba, c, hh, ww = imgs1.shape
    assert hh == ww == cfg.img and c == cfg.chans, (imgs1.shape, cfg.img, cfg.chans)
    T, L, D, Dd, p = cfg.T, cfg.L, cfg.dim, cfg.ddim, cfg.patch
    if T > 1:
        assert imgs_next.dim() == 5 and imgs_next.shape[1] == T - 1, imgs_next.shape   # multiagent_mae.py:959
    n_frames = ba * T
    random_mode = cfg.mask == "random"
    save = [] if train else None

    # ---- mask index build (K5) --------------------------------------------------------------------------
    if random_mode:
        keep = int(L * (1 - mask_ratio))                                  # mae_base.py:415
        nrows = n_frames
    else:
        keep = math.ceil(L * (1 - (1.0 - 1.0 / T)))                       # multiagent_mae.py:805-810 (Q1)
        nrows = ba
    if noise is None:
        noise = torch.rand(nrows, L, device=dev)                          # Q3: always drawn
    assert tuple(noise.shape) == (nrows, L)
    ids_shuffle, ids_restore = ops.mask_sort(noise)
    src_row, dst_row, mask, fill = ops.build_rowmaps(ids_shuffle, ids_restore, ba, T, L, keep, random_mode, cfg.decode_singletemp)

    # ---- encoder (K1, K2, K3, K4) -------------------------------------------------------------------------
    channels_last = imgs1.stride(1) == 1 and c > 1
    # BF16X3: the patches are 0/1 occupancy values, exact in plain bf16 -- only the weight is a hi/lo pair (GEMM_A_PLAIN)
    a_pe = ops.patchify(imgs1, imgs_next if T > 1 else None, T, p, BF16 if dt == BF16X3 else dt, channels_last_order=channels_last)
    pe_flags = GEMM_A_PLAIN if dt == BF16X3 else 0
    w_pe = _patch_weight(P, W, cfg, dt, channels_last)
    pos = P["pos_embed"][0, 1:]
    if cfg.encode_partial:
        _, x, _ = ops.gemm(a_pe, w_pe, bias=P["patch_embed.proj.bias"], res=pos, res_mod=L, a_rows=src_row, want_lp=False,
                           want_f32=True, flags=GEMM_RES_BY_AROW | pe_flags, dt=dt)
        l_enc = keep
    else:
        _, x, _ = ops.gemm(a_pe, w_pe, bias=P["patch_embed.proj.bias"], res=pos, res_mod=L, want_lp=False, want_f32=True, flags=pe_flags, dt=dt)
        l_enc = L
    if taps is not None:
        taps["x0"] = x.clone()
    x, y, mean_e, rstd_e = _blocks_fwd(x, P, W, "blocks.%d.", cfg.depth, "norm", cfg.heads, n_frames, l_enc, dt, cfg.eps, save, train,
                                       taps, "enc_blk%d")
    if taps is not None:
        taps["enc_norm"] = y.clone()
    comp_rows = None if cfg.encode_partial else src_row
    latent_lp, latent, _ = ops.gemm(y, _lin(P, W, "compressor.0", dt), m=n_frames * keep, bias=P["compressor.0.bias"],
                                    act=ACT_RELU, a_rows=comp_rows, want_f32=True, dt=dt)

    # ---- vector quantiser between compressor and decompressor (VQ-STAR only; VQSTAR.py:432,456) -------------------
    vq = None
    dec_in_lp = latent_lp
    if cfg.vq:
        codebook = P["_vq_star._embedding.weight"]
        vq = ops.vq_assign(latent, codebook, dt, want_f32=taps is not None)
        dec_in_lp = vq["q_lp"]
        if taps is not None:
            taps["quantized"], taps["vq_indices"] = vq["q_f32"], vq["idx"]
        if cfg.vq_decay > 0.0 and vq_update:
            # EMA codebook (mae_base.py:228-241): the quantised rows above used the codebook BEFORE this update, and so
            # does the backward pass (a copy is kept for it).  Data-parallel ranks pool their statistics first.
            vq["codebook_used"] = codebook.detach().clone() if train else None
            counts = vq["counts"]
            dw = ops.vq_code_sums(latent, vq["idx"], codebook.shape[0])
            if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
                counts = counts.clone()
                torch.distributed.all_reduce(counts)
                torch.distributed.all_reduce(dw)
            ops.vq_ema_update(counts, dw, codebook.data, P["_vq_star._ema_w"].data, P["_vq_star._ema_cluster_size"], cfg.vq_decay,
                              cfg.vq_epsilon)

    # ---- decoder ----------------------------------------------------------------------------------------------
    z, _, _ = ops.gemm(dec_in_lp, _lin(P, W, "decompressor", dt), bias=P["decompressor.bias"], dt=dt)
    dpos = P["decoder_pos_embed"][0, 1:]
    temp = None if cfg.no_temp_emb else P["decoder_temp_embed"].reshape(T, Dd)
    tok = P["mask_token"].reshape(Dd)
    w_de = _lin(P, W, "decoder_embed", dt)
    if not random_mode:
        xd = torch.empty(ba * L, Dd, dtype=torch.float32, device=dev)
        ops.gemm(z, w_de, bias=P["decoder_embed.bias"], c_rows=dst_row, out_rows=ba * L, res=dpos, res_mod=L, res2=temp,
                 res2_div=keep, res2_mod=T, want_lp=False, want_f32=True, out_f32=xd, dt=dt)
        if cfg.decode_singletemp:      # rows no slice of timestamp 0 covers: mask token, no temporal embedding
            ops.fill_rows(xd, fill, tok, None, dpos, ba, L)
    else:
        val = torch.empty(ba * L, Dd, dtype=torch.float32, device=dev)
        ops.gemm(z, w_de, bias=P["decoder_embed.bias"], c_rows=dst_row, out_rows=ba * L, res2=temp, res2_div=keep, res2_mod=T,
                 want_lp=False, want_f32=True, out_f32=val, dt=dt)
        ops.fill_rows(val, fill, tok, None if temp is None else temp[0], None, ba, L)
        xd = ops.transpose_rows(val, dpos, ba, L, Dd)                      # Q2
        if taps is not None:
            taps["dec_val"] = val.clone()
    if taps is not None:
        taps["dec_x0"] = xd.clone()
    xd, yd, mean_d, rstd_d = _blocks_fwd(xd, P, W, "decoder_blocks.%d.", cfg.ddepth, "decoder_norm", cfg.dheads, ba, L, dt, cfg.eps, save,
                                         train, taps, "dec_blk%d")
    if taps is not None:
        taps["dec_norm"] = yd.clone()
    _, logits, _ = ops.gemm(yd, _heads(P, W, cfg, dt), bias=_heads_bias(P, W, cfg, dt), want_lp=False, want_f32=True, dt=dt)

    # ---- loss, decision, fusion (K9, K10) -----------------------------------------------------------------------
    ce = ops.ce_argmax(logits, imgs1, p, BF16 if dt == BF16X3 else dt, want_pred=taps is not None, want_dlogits=train)
    if taps is not None:
        taps["pred"], taps["mask"], taps["ids_restore"] = ce["pred"], mask, ids_restore
    if theta is None:    # a captured graph brings the pose table in a static device buffer instead (star_amd/graph.py)
        theta = ops.warp_thetas(trans_matrices).pin_memory().to(dev, non_blocking=True)   # pinned: the copy does not block the host
    na = num_agent if num_agent.is_cuda else num_agent.to(dev)
    if na.dtype not in (torch.int32, torch.int64):
        na = na.long()
    result = ops.warp_fuse(ce["ind8"], theta, na.contiguous(), batch, ego=imgs1 if infer else None, binary=True)
    out = dict(loss=ce["loss"], result=result, latent=latent.reshape(n_frames, keep, cfg.inter), ind=ce["ind64"], mask=mask)
    if cfg.vq:   # loss = vq_loss + recon_loss (VQSTAR.py:443); the sum is formed on the device, in stream order
        stats = ops.vq_finish(vq["part"], vq["counts"], n_frames * keep, cfg.inter, cfg.commitment, loss_inout=ce["loss"],
                              ema=cfg.vq_decay > 0.0)
        out.update(vq_loss=stats[0:1], perplexity=stats[1:2], vq_indices=vq["idx"])
    ctx = None
    if train:
        ctx = dict(save=save, a_pe=a_pe, channels_last=channels_last, src_row=src_row, dst_row=dst_row, fill=fill, x_enc=x,
                   mean_e=mean_e, rstd_e=rstd_e, y=y, latent_lp=latent_lp, z=z, xd=xd, mean_d=mean_d, rstd_d=rstd_d, yd=yd,
                   dlogits=ce["dlogits"], keep=keep, ba=ba, l_enc=l_enc, random_mode=random_mode, dec_in_lp=dec_in_lp,
                   latent=latent if cfg.vq else None, vq_idx=vq["idx"] if cfg.vq else None,
                   vq_codebook=vq.get("codebook_used") if cfg.vq else None)
    return out, ctx


def backward(cfg, P, W, dt, ctx, G=None, on_ready=None):
    """Gradients of the mean cross-entropy w.r.t. every trainable parameter, keyed by state-dict name.
    Parameters the path never touches (cls_token, temp_embed; Q5) and the frozen tables are absent.
    G: optional dict name -> float32 view to write each gradient into (a flat gradient buffer).
    on_ready: optional callback(k) fired when the gradients of stages(cfg)[k] are complete (in stream
    order), so that a data-parallel wrapper can reduce finished buckets while the rest still runs."""
    T, L, D, Dd = cfg.T, cfg.L, cfg.dim, cfg.ddim
    ba, keep = ctx["ba"], ctx["keep"]
    n_frames = ba * T
    grads = {}
    save = ctx["save"]
    dlog = ctx["dlogits"]
    side = _SideStream(dlog.device) if SIDE_STREAM and G is not None else None
    fold = _LnFolds(W) if G is not None and LN_FOLD_BATCH else None      # flat gradient buffer: all LayerNorm folds in one launch at the end
    # prediction heads: weight gradient (and its row permutation back to the reference order) beside the dgrad chain
    side_x = side if SIDE_EXTRA else None
    queue = _WgQueue(side) if WG_GROUP and dt == BF16 else None      # grouped launches (with or without a side stream)

    def end_stage(k, sd):
        """stage k's gradients are all enqueued: hand the stage to the data-parallel wrapper.  With a side stream the hand-over
        runs ON the side stream (behind the stage's weight-gradient launches, and behind everything the main chain has enqueued
        so far): the collective then waits for that stream, and the main chain is never made to wait for the weight gradients in
        the middle of the backward (STAR_AMD_DP_JOIN=1: the round-2 behaviour, main joins the side stream at every stage)."""
        if on_ready is None:
            return
        if queue is not None:
            queue.flush()
        if sd is None:
            on_ready(k)
        elif _DP_JOIN:
            sd.join()
            on_ready(k)
        else:
            sd.stream.wait_stream(torch.cuda.current_stream(sd.device))
            with torch.cuda.stream(sd.stream):
                on_ready(k)
            sd.used = True

    def heads_unpermute(dw, db):
        idx = head_permutation(cfg, dw.device, as_int64=True)
        V = cfg.V
        for name, lo in (("decoder_pred_free", 0), ("decoder_pred_occ", V)):     # rows back in the reference order, straight into G
            gw = G[name + ".weight"] if G is not None else torch.empty(V, Dd, dtype=torch.float32, device=dw.device)
            gb = G[name + ".bias"] if G is not None else torch.empty(V, dtype=torch.float32, device=dw.device)
            gw.index_copy_(0, idx, dw[lo:lo + V])
            gb.index_copy_(0, idx, db[lo:lo + V])
            grads[name + ".weight"], grads[name + ".bias"] = gw, gb

    def heads_wgrad():
        heads_unpermute(*ops.gemm_wgrad(dlog, ctx["yd"]))
    if queue is not None and ops.wgrad_group_eligible(dlog, ctx["yd"]):
        hw = torch.empty(2 * cfg.V, Dd, dtype=torch.float32, device=dlog.device)
        hb = torch.empty(2 * cfg.V, dtype=torch.float32, device=dlog.device)
        queue.add(dlog, ctx["yd"], hw, hb, post=lambda: heads_unpermute(hw, hb))
        if side is not None:
            hw.record_stream(side.stream)
            hb.record_stream(side.stream)
    elif side_x is None:
        heads_wgrad()
    else:
        side_x.run(heads_wgrad, dlog, ctx["yd"])
    dyd, _, _ = ops.gemm(dlog, _heads(P, W, cfg, dt, True))
    g, g_lp = _ln_bwd(grads, G, "decoder_norm", dyd, ctx["xd"], P, ctx["mean_d"], ctx["rstd_d"], fold=fold)
    stage = 0
    end_stage(stage, side_x if queue is None else side)
    for s in reversed(save[cfg.depth:]):
        g, g_lp = _block_bwd(g, g_lp, s, P, W, dt, grads, G, side, fold, queue)
        stage += 1
        end_stage(stage, side)
    # temporal un-mixing backward: g is d/d(decoder input rows)
    if ctx["random_mode"]:
        g = ops.transpose_rows_bwd(g, ba, L, Dd)                                    # Q2 backward
    fill, dst_row = ctx["fill"], ctx["dst_row"]
    if ctx["random_mode"] or cfg.decode_singletemp:
        gtok = ops.colsum_groups(g, 1, 1, sel=fill)                                 # rows that took the mask token
        _put(grads, G, "mask_token", gtok.reshape(1, 1, Dd))
    _, dye = ops.gather_rows(g, dst_row, dt=dt, want_f32=False)                     # [n_frames*keep, Dd]
    if not cfg.no_temp_emb:
        gt = ops.colsum_groups(dye, keep, T)
        if ctx["random_mode"]:
            gt[0] += gtok[0]                                                        # mask tokens carry temp_embed[0]
        _put(grads, G, "decoder_temp_embed", gt.reshape(1, T, 1, Dd))
    _wg(grads, G, "decoder_embed", dye, ctx["z"], side=side_x, queue=queue)
    dz, _, _ = ops.gemm(dye, _lin(P, W, "decoder_embed", dt, True))
    _wg(grads, G, "decompressor", dz, ctx["dec_in_lp"], side=side_x, queue=queue)
    if cfg.vq:
        # straight-through estimator: the decoder's gradient passes to the latent, plus the commitment term; the codebook
        # only sees the q_latent term (mae_base.py:162-166); then the compressor's ReLU
        dq, _, _ = ops.gemm(dz, _lin(P, W, "decompressor", dt, True))
        name = "_vq_star._embedding.weight"
        if cfg.vq_decay > 0.0:      # EMA codebook: no gradient to the codes; q is read from the codebook the forward used
            used = ctx["vq_codebook"] if ctx["vq_codebook"] is not None else P[name]
            dlat, _ = ops.vq_bwd(dq, ctx["latent"], ctx["vq_idx"], used, cfg.commitment, want_dcodebook=False)
        else:
            dlat, dcb = ops.vq_bwd(dq, ctx["latent"], ctx["vq_idx"], P[name], cfg.commitment, dcodebook=None if G is None else G[name])
            grads[name] = dcb
    else:
        dlat, _, _ = ops.gemm(dz, _lin(P, W, "decompressor", dt, True), act=ACT_DRELU, aux=ctx["latent_lp"])
    comp_rows = None if cfg.encode_partial else ctx["src_row"]
    _wg(grads, G, "compressor.0", dlat, ctx["y"], x_rows=comp_rows, side=side_x, queue=queue)
    rows_enc = n_frames * ctx["l_enc"]
    dy, _, _ = ops.gemm(dlat, _lin(P, W, "compressor.0", dt, True), c_rows=comp_rows, out_rows=rows_enc)
    g, g_lp = _ln_bwd(grads, G, "norm", dy, ctx["x_enc"], P, ctx["mean_e"], ctx["rstd_e"], fold=fold)
    stage += 1
    end_stage(stage, side_x if queue is None else side)
    for s in reversed(save[:cfg.depth]):
        g, g_lp = _block_bwd(g, g_lp, s, P, W, dt, grads, G, side, fold, queue)
        stage += 1
        end_stage(stage, side)
    # patch embedding
    wshape = P["patch_embed.proj.weight"].shape
    wname, bname = "patch_embed.proj.weight", "patch_embed.proj.bias"
    # written in place where the layouts allow it: the weight gradient unless its columns come in (py, px, c) order, the bias always
    dw_out = G[wname].view(wshape[0], -1) if (G is not None and not ctx["channels_last"]) else None
    if cfg.encode_partial:
        dw, _ = ops.gemm_wgrad(g_lp, ctx["a_pe"], x_rows=ctx["src_row"], want_bias=False, dw=dw_out)
        db = ops.colsum_groups(g, 1, 1, sel=ctx["src_row"], sel_nonneg=True).reshape(-1)   # padded slots carry no bias
        _put(grads, G, bname, db)
    elif queue is not None and G is not None and ops.wgrad_group_eligible(g_lp, ctx["a_pe"]):
        if dw_out is not None:
            dw, grads[bname] = queue.add(g_lp, ctx["a_pe"], dw_out, G[bname])       # rides with the last block's products
        else:       # channels-last patches: the columns come out in (py, px, c) order, one strided copy behind the launch puts them into G
            tmp = torch.empty(wshape[0], wshape[1] * wshape[2] * wshape[3], dtype=torch.float32, device=g_lp.device)
            if side is not None:
                tmp.record_stream(side.stream)

            def permute_into_g():
                G[wname].copy_(tmp.reshape(wshape[0], wshape[2], wshape[3], wshape[1]).permute(0, 3, 1, 2))
            _, grads[bname] = queue.add(g_lp, ctx["a_pe"], tmp, G[bname], post=permute_into_g)
            grads[wname] = G[wname]
            dw = None
    else:
        dw, db = ops.gemm_wgrad(g_lp, ctx["a_pe"], dw=dw_out, db=None if G is None else G[bname])
        grads[bname] = db
    if dw is None:
        pass                                                                               # queued: the copy into G runs behind the grouped launch
    elif ctx["channels_last"]:
        dw = dw.reshape(wshape[0], wshape[2], wshape[3], wshape[1]).permute(0, 3, 1, 2)    # strided view: ONE copy into G (or below)
        if G is None:
            dw = dw.contiguous()
        _put(grads, G, wname, dw)
    else:
        grads[wname] = G[wname] if G is not None else dw.reshape(wshape)
    if queue is not None:
        # the patch embedding's product is the last of the backward and the main chain has nothing left to run: launched on the main stream
        # it overlaps the first block's grouped launch (still running on the side stream) instead of queueing behind it
        if _WG_TAIL_MAIN:
            queue.side = None
        queue.flush()
    if fold is not None:
        fold.run()
    if side is not None:
        side.join()
    if on_ready is not None:
        on_ready(stage + 1)
    return grads
