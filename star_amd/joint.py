"""Forward / backward orchestration of the JOINT (latent-level fusion) STAR variant over libstar_hip.so -- SURVEY 8(f) rank 1.

Reference: AmortizedFusionMMAEViT.forward (star/models/multiagent_mae.py:659-681): forward_encoder (:553-583) ->
forward_fusion (:497-551) -> forward_decoder (:585-605, two ConvPred heads, star/models/mae_base.py:16-105) ->
CrossEntropyLoss against the TEACHER (:646-653) -> argmax.  Returns (loss, result int64, mask, pred [BA,2,C,H,W]).

Reuses the kernels of the individual model for encoder, decoder blocks and the un-masking scatter; the conv head is
channels-last end to end: a 3x3 convolution = nine star_gemm taps whose A rows are gathered through a shifted-pixel table
(-1 = zero padding), accumulated in float32 through the residual operand; its data gradient = the same taps with the opposite
shifts on the transposed weights; its weight gradient = nine star_gemm_wgrad taps.

Reference behaviour kept on purpose: fusion("sum") REPLACES the fused latent by {0, 1} through masked assignment, so no
gradient reaches the encoder through a fused slot -- only agent slots >= num_agent (which keep their own features) pass one."""
import math
import os

import torch

from . import ops
from ._lib import ACT_DRELU, ACT_RELU, BF16
from .engine import WG_GROUP, _block_bwd, _blocks_fwd, _lin, _ln_bwd, _patch_weight, _put, _wg, _WgQueue

BN_EPS, BN_MOMENTUM = 1e-5, 0.1
_ROWS = {}


def conv_rows(n, h, w, device):
    """int32 [9, n*h*w]: for tap t = ky*3 + kx the row of pixel (y + ky - 1, x + kx - 1) of the same image, -1 outside."""
    key = (n, h, w, str(device))
    tab = _ROWS.get(key)
    if tab is None:
        if len(_ROWS) > 16:
            _ROWS.clear()
        yy, xx = torch.meshgrid(torch.arange(h, device=device), torch.arange(w, device=device), indexing="ij")
        base = (torch.arange(n, device=device) * (h * w)).view(n, 1, 1)
        taps = []
        for ky in range(3):
            for kx in range(3):
                y2, x2 = yy + ky - 1, xx + kx - 1
                ok = (y2 >= 0) & (y2 < h) & (x2 >= 0) & (x2 < w)
                taps.append(torch.where(ok.unsqueeze(0), base + (y2 * w + x2).unsqueeze(0), torch.full_like(base, -1)).reshape(-1))
        tab = _ROWS[key] = torch.stack(taps).to(torch.int32).contiguous()
    return tab


def _pad_to(t, n):
    if t.shape[0] == n:
        return t.contiguous()
    out = torch.zeros((n,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    out[:t.shape[0]] = t
    return out


def _conv_weights(P, W, name, dt, ldo):
    """Tap matrices of conv `name`: fwd [9, ldo, Cin] and transposed [9, Cin, ldo] in dt (output channels zero-padded to ldo)."""
    w = P[name + ".weight"]

    def build():
        w2 = _pad_to(w.detach(), ldo)                                   # [ldo, Cin, 3, 3]
        taps = w2.permute(2, 3, 0, 1).reshape(9, ldo, w.shape[1]).contiguous()
        return ops.cast(taps, dt), ops.cast(taps.transpose(1, 2).contiguous(), dt)

    return W.get(("conv", name, dt, ldo), w._version, build)


_CONV_FUSED = os.environ.get("STAR_AMD_CONV_FUSED", "1") != "0"     # 0: nine star_gemm taps through the float32 output (A/B)


def _conv_fused_ok(dt, h, w, kin):
    """star_gemm's implicit 3 x 3 convolution covers it: bf16, power-of-two images, input channels a multiple of 32"""
    return _CONV_FUSED and dt == BF16 and h & (h - 1) == 0 and w & (w - 1) == 0 and kin % 32 == 0


def _conv_weights_fused(P, W, name, dt, ldo):
    """The tap matrices side by side: forward [ldo, 9 Cin] (tap t in columns t Cin ..) and, for the data gradient, [Cin, 9 ldo] with
    the taps REVERSED (input pixel p collects from output pixel p - shift(t): the opposite tap's shift)."""
    w = P[name + ".weight"]

    def build():
        taps, taps_t = _conv_weights(P, W, name, dt, ldo)
        return (taps.permute(1, 0, 2).reshape(ldo, -1).contiguous(), taps_t.flip(0).permute(1, 0, 2).reshape(taps_t.shape[1], -1).contiguous())

    return W.get(("conv_fused", name, dt, ldo), w._version, build)


def _conv3x3(x, taps, bias, rows, ldo, fused=None, hw=None):
    """x dt [R, Cin] -> float32 [R, ldo].  fused: the [ldo, 9 Cin] weight -> one launch (the taps accumulate in registers)."""
    out = torch.empty(x.shape[0], ldo, dtype=torch.float32, device=x.device)
    if fused is not None:
        ops.gemm(x, fused, bias=bias, want_lp=False, want_f32=True, out_f32=out, conv=hw)
        return out
    for t in range(9):
        ops.gemm(x, taps[t], bias=bias if t == 0 else None, res=out if t > 0 else None, a_rows=rows[t], want_lp=False, want_f32=True,
                 out_f32=out)
    return out


def _stages(P, pre):
    s = 1
    while (pre + "conv%d.weight" % (s + 1)) in P:
        s += 1
    return s


def head_forward(P, W, pre, x, n, g, dt, training, save):
    """ConvPred / ConvPred16 forward (mae_base.py:37-54, 82-105) on channels-last rows.  -> float32 logits [n*H*W, ld]."""
    S = _stages(P, pre)
    h = w = g
    for s in range(1, S + 1):
        cout = P[pre + "conv%d.weight" % s].shape[0]
        ldo = (cout + 7) // 8 * 8
        xu = ops.upsample2x(x, n, h, w)
        h, w = 2 * h, 2 * w
        rows = conv_rows(n, h, w, x.device)
        taps, _ = _conv_weights(P, W, pre + "conv%d" % s, dt, ldo)
        fused = _conv_weights_fused(P, W, pre + "conv%d" % s, dt, ldo)[0] if _conv_fused_ok(dt, h, w, xu.shape[1]) else None
        conv = _conv3x3(xu, taps, _pad_to(P[pre + "conv%d.bias" % s].detach(), ldo), rows, ldo, fused=fused, hw=(h, w))
        gamma, beta = _pad_to(P[pre + "bn%d.weight" % s].detach(), ldo), _pad_to(P[pre + "bn%d.bias" % s].detach(), ldo)
        rm, rv = P[pre + "bn%d.running_mean" % s], P[pre + "bn%d.running_var" % s]
        if training:
            mean, var, rstd = ops.bn_stats(conv, BN_EPS)
            r = conv.shape[0]
            with torch.no_grad():       # nn.BatchNorm2d's running statistics (unbiased variance), momentum 0.1
                rm.mul_(1 - BN_MOMENTUM).add_(mean[:cout], alpha=BN_MOMENTUM)
                rv.mul_(1 - BN_MOMENTUM).add_(var[:cout] * (r / max(r - 1, 1)), alpha=BN_MOMENTUM)
                P[pre + "bn%d.num_batches_tracked" % s].add_(1)
        else:
            mean = _pad_to(rm.detach(), ldo)
            rstd = _pad_to(torch.rsqrt(rv.detach() + BN_EPS), ldo)
        last = s == S
        y = ops.bn_apply(conv, mean, rstd, gamma, beta, relu=not last, dt=1 if last else dt)     # logits stay float32
        if save is not None:
            save.append(dict(pre=pre, s=s, xu=xu, conv=conv, mean=mean, rstd=rstd, gamma=gamma, beta=beta, h=h, w=w, cout=cout, ldo=ldo, last=last))
        x = y
    return x


def head_backward(P, W, saved, d, n, dt, training, grads):
    """d: float32 gradient of the head's logits [R, ld].  -> float32 gradient of the head's input [n*g*g, C0]."""
    for sv in reversed(saved):
        pre, s, cout, ldo = sv["pre"], sv["s"], sv["cout"], sv["ldo"]
        dconv, dgamma, dbeta = ops.bn_bwd(sv["conv"], d, sv["mean"], sv["rstd"], sv["gamma"], sv["beta"], relu=not sv["last"], training=training, dt=dt)
        grads[pre + "bn%d.weight" % s], grads[pre + "bn%d.bias" % s] = dgamma[:cout].clone(), dbeta[:cout].clone()
        rows = conv_rows(n, sv["h"], sv["w"], d.device)
        cin = sv["xu"].shape[1]
        hh, ww = sv["h"], sv["w"]
        if (dt == ops.BF16 and hh & (hh - 1) == 0 and ww & (ww - 1) == 0 and ops.wgrad_group_eligible(dconv, sv["xu"])
                and os.environ.get("STAR_AMD_CONV_WGRAD_GROUP", "1") != "0"):
            # the nine taps as grouped launches: the shifted-pixel gather is arithmetic inside the kernel (no row table)
            dwt = torch.empty(9, ldo, cin, dtype=torch.float32, device=d.device)
            db = torch.empty(ldo, dtype=torch.float32, device=d.device)
            ops.gemm_wgrad_group([(dconv, sv["xu"], dwt[t], db if t == 0 else False, ops.wgrad_tap(t // 3 - 1, t % 3 - 1, hh, ww)) for t in range(9)])
            dws = list(dwt)
            grads[pre + "conv%d.bias" % s] = db[:cout].clone()
        else:
            dws = []
            for t in range(9):
                dw, db = ops.gemm_wgrad(dconv, sv["xu"], x_rows=rows[t], want_bias=(t == 0))
                dws.append(dw)
                if t == 0:
                    grads[pre + "conv%d.bias" % s] = db[:cout].clone()
        grads[pre + "conv%d.weight" % s] = torch.stack(dws).reshape(3, 3, ldo, cin)[:, :, :cout].permute(2, 3, 0, 1).contiguous()
        _, tapsT = _conv_weights(P, W, pre + "conv%d" % s, dt, ldo)
        dxu = torch.empty(dconv.shape[0], cin, dtype=torch.float32, device=d.device)
        if _conv_fused_ok(dt, hh, ww, ldo) and cin % 16 == 0:
            ops.gemm(dconv, _conv_weights_fused(P, W, pre + "conv%d" % s, dt, ldo)[1], want_lp=False, want_f32=True, out_f32=dxu, conv=(hh, ww))
        else:
            for t in range(9):                          # input pixel p collects from output pixel p - shift(t): the opposite tap's table
                ops.gemm(dconv, tapsT[t], res=dxu if t > 0 else None, a_rows=rows[8 - t], want_lp=False, want_f32=True, out_f32=dxu)
        d = ops.upsample2x_bwd(dxu, n, sv["h"] // 2, sv["w"] // 2)
    return d


# ----------------------------------------------------------------------------------------------------------------------
def forward(cfg, P, W, dt, imgs1, imgs_next, teacher, trans_matrices, num_agent, batch, mask_ratio, train, training, noise=None, taps=None):
    """-> (out dict(loss [1], result int64, mask, pred), ctx for backward).  train: keep activations; training: BatchNorm mode."""
    dev = imgs1.device
    ba, c, hh, ww = imgs1.shape
    assert hh == ww == cfg.img and c == cfg.chans, (imgs1.shape, cfg.img, cfg.chans)
    T, L, D, Dd, p, G = cfg.T, cfg.L, cfg.dim, cfg.ddim, cfg.patch, cfg.grid
    if T > 1:
        assert imgs_next.dim() == 5 and imgs_next.shape[1] == T - 1, imgs_next.shape     # multiagent_mae.py:566
    n_frames = ba * T
    random_mode = cfg.mask == "random"
    save = [] if train else None
    if random_mode:
        keep, nrows = int(L * (1 - mask_ratio)), n_frames
    else:
        keep, nrows = math.ceil(L * (1 - (1.0 - 1.0 / T))), ba
    if noise is None:
        noise = torch.rand(nrows, L, device=dev)
    assert tuple(noise.shape) == (nrows, L)
    ids_shuffle, ids_restore = ops.mask_sort(noise)
    src_row, dst_row, mask, fill = ops.build_rowmaps(ids_shuffle, ids_restore, ba, T, L, keep, random_mode, False)

    # ---- encoder: every token of every timestamp, mask only what is transmitted (multiagent_mae.py:553-583) --------------
    channels_last = imgs1.stride(1) == 1 and c > 1
    a_pe = ops.patchify(imgs1, imgs_next if T > 1 else None, T, p, dt, channels_last_order=channels_last)
    _, x, _ = ops.gemm(a_pe, _patch_weight(P, W, cfg, dt, channels_last), bias=P["patch_embed.proj.bias"], res=P["pos_embed"][0, 1:], res_mod=L,
                       want_lp=False, want_f32=True)
    x, y, mean_e, rstd_e = _blocks_fwd(x, P, W, "blocks.%d.", cfg.depth, "norm", cfg.heads, n_frames, L, dt, cfg.eps, save, train)
    latent_lp, _, _ = ops.gemm(y, _lin(P, W, "compressor.0", dt), m=n_frames * keep, bias=P["compressor.0.bias"], act=ACT_RELU, a_rows=src_row)

    # ---- forward_fusion (:497-551): decompress, embed, un-mask (+ temporal embedding), warp + sum + threshold ---------------
    z, _, _ = ops.gemm(latent_lp, _lin(P, W, "decompressor", dt), bias=P["decompressor.bias"])
    temp = P["decoder_temp_embed"].reshape(T, Dd)
    feat = torch.empty(ba * L, Dd, dtype=torch.float32, device=dev)
    ops.gemm(z, _lin(P, W, "decoder_embed", dt), bias=P["decoder_embed.bias"], c_rows=dst_row, out_rows=ba * L, res2=temp, res2_div=keep, res2_mod=T,
             want_lp=False, want_f32=True, out_f32=feat)
    if random_mode:
        ops.fill_rows(feat, fill, P["mask_token"].reshape(Dd), temp[0], None, ba, L)
    theta = ops.warp_thetas(trans_matrices).pin_memory().to(dev, non_blocking=True)
    na = num_agent if num_agent.is_cuda else num_agent.to(dev)
    if na.dtype not in (torch.int32, torch.int64):
        na = na.long()
    xd, passthrough = ops.feat_warp_fuse(feat.reshape(ba, L, Dd), theta, na.contiguous(), batch, G, pos=P["decoder_pos_embed"][0, 1:].contiguous())
    if taps is not None:
        taps["feat"], taps["fused"] = feat.clone(), xd.clone()
    xd = xd.reshape(ba * L, Dd)

    # ---- forward_decoder (:585-605) -----------------------------------------------------------------------------------
    xd, yd, mean_d, rstd_d = _blocks_fwd(xd, P, W, "decoder_blocks.%d.", cfg.ddepth, "decoder_norm", cfg.dheads, ba, L, dt, cfg.eps, save, train)
    if taps is not None:
        taps["dec_norm"] = yd.clone()
    hs_free, hs_occ = ([] if train else None), ([] if train else None)
    lfree = head_forward(P, W, "decoder_pred_free.", yd, ba, G, dt, training, hs_free)
    locc = head_forward(P, W, "decoder_pred_occ.", yd, ba, G, dt, training, hs_occ)
    ce = ops.ce_nhwc(lfree, locc, teacher, c, want_pred=True, want_grad=train)
    out = dict(loss=ce["loss"], result=ce["ind64"], mask=mask, pred=ce["pred"])
    ctx = None
    if train:
        ctx = dict(save=save, a_pe=a_pe, channels_last=channels_last, src_row=src_row, dst_row=dst_row, fill=fill, x_enc=x, mean_e=mean_e,
                   rstd_e=rstd_e, y=y, latent_lp=latent_lp, z=z, xd=xd, mean_d=mean_d, rstd_d=rstd_d, hs_free=hs_free, hs_occ=hs_occ,
                   dfree=ce["dfree"], docc=ce["docc"], keep=keep, ba=ba, random_mode=random_mode, passthrough=passthrough, training=training)
    return out, ctx


def backward(cfg, P, W, dt, ctx):
    """Gradients of the loss w.r.t. every parameter the path touches, keyed by state-dict name."""
    T, L, Dd = cfg.T, cfg.L, cfg.ddim
    ba, keep = ctx["ba"], ctx["keep"]
    n_frames = ba * T
    grads = {}
    save = ctx["save"]
    # weight gradients of the Linears leave in grouped launches (star_gemm_wgrad_group), one per block, as in the individual model
    queue = _WgQueue(None) if WG_GROUP else None
    d_yd = head_backward(P, W, ctx["hs_free"], ctx["dfree"], ba, dt, ctx["training"], grads)
    d_yd = d_yd + head_backward(P, W, ctx["hs_occ"], ctx["docc"], ba, dt, ctx["training"], grads)
    d_lp = d_yd if dt != BF16 else ops.cast(d_yd, dt)
    g, g_lp = _ln_bwd(grads, None, "decoder_norm", d_lp, ctx["xd"], P, ctx["mean_d"], ctx["rstd_d"])
    for s in reversed(save[cfg.depth:]):
        g, g_lp = _block_bwd(g, g_lp, s, P, W, dt, grads, queue=queue)
    # forward_fusion backward: the thresholded rows are constants; slots that kept their own features pass the gradient on
    pt_rows = ctx["passthrough"].to(torch.bool).repeat_interleave(L)                 # [ba*L] index bookkeeping, no arithmetic
    dst_row, fill = ctx["dst_row"], ctx["fill"]
    dst_pt = torch.where((dst_row >= 0) & pt_rows[dst_row.clamp(min=0).long()], dst_row, torch.full_like(dst_row, -1))
    if ctx["random_mode"]:
        fill_pt = torch.where(pt_rows, fill, torch.zeros_like(fill))
        gtok = ops.colsum_groups(g, 1, 1, sel=fill_pt)
        _put(grads, None, "mask_token", gtok.reshape(1, 1, Dd))
    _, dye = ops.gather_rows(g, dst_pt, dt=dt, want_f32=False)
    gt = ops.colsum_groups(dye, keep, T)
    if ctx["random_mode"]:
        gt[0] += gtok[0]
    _put(grads, None, "decoder_temp_embed", gt.reshape(1, T, 1, Dd))
    _wg(grads, None, "decoder_embed", dye, ctx["z"], queue=queue)
    dz, _, _ = ops.gemm(dye, _lin(P, W, "decoder_embed", dt, True))
    _wg(grads, None, "decompressor", dz, ctx["latent_lp"], queue=queue)
    dlat, _, _ = ops.gemm(dz, _lin(P, W, "decompressor", dt, True), act=ACT_DRELU, aux=ctx["latent_lp"])
    _wg(grads, None, "compressor.0", dlat, ctx["y"], x_rows=ctx["src_row"], queue=queue)
    if queue is not None:
        queue.flush()
    dy, _, _ = ops.gemm(dlat, _lin(P, W, "compressor.0", dt, True), c_rows=ctx["src_row"], out_rows=n_frames * L)
    g, g_lp = _ln_bwd(grads, None, "norm", dy, ctx["x_enc"], P, ctx["mean_e"], ctx["rstd_e"])
    for s in reversed(save[:cfg.depth]):
        g, g_lp = _block_bwd(g, g_lp, s, P, W, dt, grads, queue=queue)
    if queue is not None:
        queue.flush()
    wshape = P["patch_embed.proj.weight"].shape
    dw, db = ops.gemm_wgrad(g_lp, ctx["a_pe"])
    if ctx["channels_last"]:
        dw = dw.reshape(wshape[0], wshape[2], wshape[3], wshape[1]).permute(0, 3, 1, 2).contiguous()
    grads["patch_embed.proj.weight"], grads["patch_embed.proj.bias"] = dw.reshape(wshape), db
    return grads
