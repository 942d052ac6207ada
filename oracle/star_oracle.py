"""CPU ORACLE for the STAR completion hot path.  TEST INFRASTRUCTURE ONLY.

This file restates, in plain PyTorch-CPU / numpy, the algorithm of the reference path that
`star_amd` accelerates.  Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline`
leg may import it; the product (`star_amd/`) never does, and fails loudly when its HIP library is
missing instead of falling back to anything here.

Parity status: PINNED.  Every function below is checked (tests/test_oracle_golden.py) against golden
vectors captured from the reference's own model files imported in the build container
(tests/golden/gen_golden.py).  The transformer block / patch embedding arithmetic lives in the
third-party dependency timm==0.3.2 (reference requirements.txt:14), absent from the container; its
published 0.3.2 definition is restated in `block()` / `patch_embed()` and the goldens pin the
composite "reference model files + that definition".

It is functional on purpose (a state dict of tensors + a config dict -> outputs): nothing here is an
nn.Module, so nothing can be mistaken for, or reused as, the product's model class.

Reference files followed (paths under /root/reference):
  star/models/mae_base.py:257-750, star/models/multiagent_mae.py:684-1141,
  star/utils/pos_embed.py:20-67, star/utils/lr_sched.py:9-21, star/utils/metrics.py:14-82,128-218,
  star/utils/misc.py:251-292, star/utils/CoModule.py:91-156.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

F32 = np.float32


# --------------------------------------------------------------------------------------------
# configuration
# --------------------------------------------------------------------------------------------
def make_cfg(img=256, patch=8, chans=13, dim=384, depth=6, heads=12, ddim=256, ddepth=4, dheads=16,
             mlp_ratio=4, T=1, mask="complement", encode_partial=False, no_temp_emb=False,
             decode_singletemp=False, eps=1e-6, inter=32):
    """Hyper-parameters of one AmortizedIndivMMAEViT instance (multiagent_mae.py:688-739,1129-1184)."""
    if mask not in ("random", "complement"):
        raise NotImplementedError(mask)            # multiagent_mae.py:727
    return dict(img=img, patch=patch, chans=chans, dim=dim, depth=depth, heads=heads, ddim=ddim,
                ddepth=ddepth, dheads=dheads, mlp_ratio=mlp_ratio, T=T, mask=mask,
                encode_partial=encode_partial, no_temp_emb=no_temp_emb,
                decode_singletemp=decode_singletemp, eps=eps, inter=inter)


# --------------------------------------------------------------------------------------------
# host-side tables and schedules
# --------------------------------------------------------------------------------------------
def sincos_1d(dim, pos):
    """pos_embed.py:49-67: [sin | cos] of pos * 10000^(-i/(dim/2)), float64."""
    omega = 1.0 / 10000 ** (np.arange(dim // 2, dtype=np.float64) / (dim / 2.0))
    ang = np.outer(pos.reshape(-1).astype(np.float64), omega)
    return np.concatenate([np.sin(ang), np.cos(ang)], axis=1)


def sincos_pos_embed(dim, grid, cls_token=True):
    """pos_embed.py:20-46.  First half of the channels encodes the COLUMN index (the reference's
    meshgrid puts w first, :28), second half the row index; row 0 is the all-zero cls slot."""
    cols, rows = np.meshgrid(np.arange(grid, dtype=np.float32), np.arange(grid, dtype=np.float32))
    emb = np.concatenate([sincos_1d(dim // 2, cols), sincos_1d(dim // 2, rows)], axis=1)
    if cls_token:
        emb = np.concatenate([np.zeros((1, dim)), emb], axis=0)
    return emb


def lr_at(epoch, lr, min_lr, warmup_epochs, nepoch):
    """lr_sched.py:9-21: linear warm-up then half cosine; `epoch` is fractional."""
    if epoch < warmup_epochs:
        return lr * epoch / warmup_epochs
    return min_lr + (lr - min_lr) * 0.5 * (1.0 + math.cos(math.pi * (epoch - warmup_epochs) / (nepoch - warmup_epochs)))


def weight_decay_groups(named_params, weight_decay):
    """timm 0.3.2 optim_factory.add_weight_decay as used at train_completion.py:124: 1-D tensors and
    biases get no decay; frozen tensors are skipped."""
    decay, no_decay = [], []
    for name, p in named_params:
        if not p.requires_grad:
            continue
        (no_decay if (p.ndim == 1 or name.endswith(".bias")) else decay).append(p)
    return [{"params": no_decay, "weight_decay": 0.0}, {"params": decay, "weight_decay": weight_decay}]


# --------------------------------------------------------------------------------------------
# occupancy IoU evaluator
# --------------------------------------------------------------------------------------------
def _pool2(x):
    """metrics.py:136: MaxPool3d((2,2,2), stride 2, padding (1,0,0)) on [N,1,Z,H,W]."""
    return F.max_pool3d(x, kernel_size=2, stride=2, padding=(1, 0, 0))


def occupancy_confusions(pred, target):
    """metrics.py:146-190 + 35-52: per-scale 2x2 confusion (rows = prediction, cols = target)."""
    p1, t1 = pred.detach().float().cpu().unsqueeze(1), target.detach().float().cpu().unsqueeze(1)
    p2, t2 = _pool2(p1), _pool2(t1)
    p4, t4 = _pool2(p2), _pool2(t2)
    confs = []
    for p, t in ((p1, t1), (p2, t2), (p4, t4)):
        pi = (p >= 0.5).reshape(-1).numpy().astype(np.int64)
        ti = t.reshape(-1).numpy().astype(np.int64)      # target is cast, not thresholded (:182)
        confs.append(np.bincount(pi * 2 + ti, minlength=4).reshape(2, 2))
    return np.stack(confs)


def occupancy_iou(conf):
    """metrics.py:199-207 for one scale."""
    tp, fp, fn = conf[1:, 1:].sum(), conf[1:, 0].sum(), conf[0, 1:].sum()
    return tp / (tp + fp + fn + 1e-15)


class IoUMeter:
    """metrics.py:210-218: one IoU triple per batch, averaged over batches."""

    def __init__(self):
        self.per_batch = []

    def add_batch(self, pred, target):
        self.per_batch.append(np.array([occupancy_iou(c) for c in occupancy_confusions(pred, target)]))

    def average(self):
        return np.mean(np.asarray(self.per_batch), axis=0)


# --------------------------------------------------------------------------------------------
# patch reshapes
# --------------------------------------------------------------------------------------------
def patchify(imgs, p):
    """mae_base.py:377-390: [N,C,H,W] -> [N,L,p*p*C], vector order (py, px, c)."""
    n, c, hh, ww = imgs.shape
    assert hh == ww and hh % p == 0
    g = hh // p
    return imgs.reshape(n, c, g, p, g, p).permute(0, 2, 4, 3, 5, 1).reshape(n, g * g, p * p * c)


def unpatchify(x, p):
    """mae_base.py:392-406: inverse of patchify."""
    n, L, v = x.shape
    g = int(L ** 0.5)
    assert g * g == L
    c = v // (p * p)
    assert p * p * c == v
    return x.reshape(n, g, g, p, p, c).permute(0, 5, 1, 3, 2, 4).reshape(n, c, g * p, g * p)


# --------------------------------------------------------------------------------------------
# pose warp (nearest, zero padding) and union fusion: bit-exact integer-valued path
# --------------------------------------------------------------------------------------------
def _fma32(a, b, c):
    """float32 fused multiply-add.  The float64 product of two float32 numbers is exact; the float64
    sum and the final rounding to float32 reproduce fmaf except for double-rounding cases of
    probability ~2^-29 per operation (never seen against the reference; oracle/warp_oracle.c uses a
    true fmaf and is cross-checked against this)."""
    return (np.asarray(a, np.float64) * np.asarray(b, np.float64) + np.asarray(c, np.float64)).astype(F32)


def base_grid(n):
    """ATen linspace(-1,1,n) (symmetric two-sided fma form) scaled by (n-1)/n, all in float32:
    what F.affine_grid(..., align_corners=False) builds (mae_base.py:602,607)."""
    i = np.arange(n)
    step = F32(F32(2.0) / F32(n - 1))
    lo = _fma32(step, i.astype(F32), F32(-1.0))
    hi = _fma32(-step, (n - 1 - i).astype(F32), F32(1.0))
    ls = np.where(i < n // 2, lo, hi).astype(F32)
    return ((ls * F32(n - 1)).astype(F32) / F32(n)).astype(F32)


def affine_coords(theta, h, w):
    """base_grid [x, y, 1] times theta^T as ATen's float32 bmm evaluates it on CPU: a k-ordered fma
    chain (x*t0 rounded, then fma(y,t1,.), then fma(1,t2,.))."""
    bx = np.broadcast_to(base_grid(w)[None, :], (h, w))
    by = np.broadcast_to(base_grid(h)[:, None], (h, w))
    out = []
    for r in range(2):
        t0, t1, t2 = (F32(v) for v in theta[r])
        acc = (bx * t0).astype(F32)
        acc = _fma32(by, t1, acc)
        acc = _fma32(F32(1.0), t2, acc)
        out.append(acc)
    return out  # gx, gy


def sample_nearest(x, gx, gy):
    """grid_sample(mode='nearest', padding zeros, align_corners=False) (mae_base.py:610-611):
    un-normalise ((g+1)*size-1)/2 in float32, round half to even, zero outside."""
    c, h, w = x.shape
    ux = ((((gx + F32(1.0)).astype(F32) * F32(w)).astype(F32) - F32(1.0)).astype(F32) / F32(2.0)).astype(F32)
    uy = ((((gy + F32(1.0)).astype(F32) * F32(h)).astype(F32) - F32(1.0)).astype(F32) / F32(2.0)).astype(F32)
    ix, iy = np.rint(ux), np.rint(uy)
    ok = (ix >= 0) & (ix < w) & (iy >= 0) & (iy < h)
    ixc = np.clip(ix, 0, w - 1).astype(np.int64)
    iyc = np.clip(iy, 0, h - 1).astype(np.int64)
    return np.where(ok[None], x[:, iyc, ixc], x.dtype.type(0))


def warp_thetas(m):
    """mae_base.py:593-606: rotation then translation, entries formed in the SOURCE dtype of the
    pose matrix and only then cast to float32."""
    m = np.asarray(m)
    x_t = (4 * m[0, 3]) / 128
    y_t = -(4 * m[1, 3]) / 128
    rot = np.array([[m[0, 0], m[0, 1], 0.0], [m[1, 0], m[1, 1], 0.0]]).astype(F32)
    tr = np.array([[1.0, 0.0, x_t], [0.0, 1.0, y_t]]).astype(F32)
    return rot, tr


def warp_map(x, m):
    """feature_transformation (mae_base.py:589-614): two chained nearest resamples."""
    c, h, w = x.shape
    rot, tr = warp_thetas(m)
    gx, gy = affine_coords(rot, h, w)
    y = sample_nearest(x, gx, gy)
    gx, gy = affine_coords(tr, h, w)
    return sample_nearest(y, gx, gy)


def late_fusion(maps, trans, num_agent, batch, ego_maps=None):
    """late_fusion / ego_late_fusion (multiagent_mae.py:870-944) with fusion('union')
    (mae_base.py:624-633).  `maps` [A*B,C,H,W] agent-major; the ego slot uses `ego_maps` when given
    (inference: the agent's own input).  Slots >= num_agent keep their own map."""
    x = np.asarray(maps, dtype=F32)
    ab = x.shape[0]
    assert ab % batch == 0, (x.shape, batch)
    a = ab // batch
    out = x.copy()
    e = x if ego_maps is None else np.asarray(ego_maps, dtype=F32)
    trans = np.asarray(trans)
    for b in range(batch):
        n = int(np.asarray(num_agent)[b, 0])
        for i in range(n):
            acc = e[i * batch + b].copy()
            for j in range(n):
                if j != i:
                    acc = acc + warp_map(x[j * batch + b], trans[b, i, j])
            out[i * batch + b] = np.where(acc > 0, F32(1.0), acc)
    return out


# --------------------------------------------------------------------------------------------
# mask index build / sub-sampling / temporal mixing
# --------------------------------------------------------------------------------------------
def shuffle_ids(noise):
    """argsort twice (multiagent_mae.py:813-814, mae_base.py:420-421); stable so that ties in the
    noise resolve by position."""
    ids_shuffle = torch.argsort(noise, dim=1, stable=True)
    ids_restore = torch.argsort(ids_shuffle, dim=1, stable=True)
    return ids_shuffle, ids_restore


def take_rows(x, idx):
    return torch.gather(x, 1, idx.unsqueeze(-1).expand(-1, -1, x.shape[2]))


def complement_masking(x, T, noise=None):
    """amortized_complement_masking (multiagent_mae.py:800-834).  x [BA*T, L, D]; one noise row per
    agent-scene shared by its T timestamps; slice t of the shuffled order goes to timestamp t, the
    last slice zero-padded.  Q1: the mask ratio is 1 - 1/T whatever the caller passes."""
    bat, L, d = x.shape
    assert bat % T == 0
    ba = bat // T
    keep = math.ceil(L * (1 - (1.0 - 1.0 / T)))
    if noise is None:
        noise = torch.rand(ba, L)
    ids_shuffle, ids_restore = shuffle_ids(noise)
    xs = x.reshape(ba, T, L, d)
    mask = torch.ones(ba, T, L)
    parts = []
    for t in range(T):
        lo, hi = keep * t, min(keep * (t + 1), L)
        part = take_rows(xs[:, t], ids_shuffle[:, lo:hi])
        if hi - lo < keep:
            part = torch.cat([part, torch.zeros(ba, keep - (hi - lo), d)], dim=1)
        parts.append(part)
        m = torch.ones(ba, L)
        m[:, lo:hi] = 0
        mask[:, t] = torch.gather(m, 1, ids_restore)
    return torch.stack(parts, dim=1).reshape(bat, keep, d), mask, ids_restore


def random_masking(x, mask_ratio, noise=None):
    """amortized_random_masking -> random_masking (multiagent_mae.py:743-750, mae_base.py:408-433):
    independent noise per (agent-scene, timestamp) row."""
    n, L, d = x.shape
    keep = int(L * (1 - mask_ratio))
    if noise is None:
        noise = torch.rand(n, L)
    ids_shuffle, ids_restore = shuffle_ids(noise)
    m = torch.ones(n, L)
    m[:, :keep] = 0
    return take_rows(x, ids_shuffle[:, :keep]), torch.gather(m, 1, ids_restore), ids_restore


def complement_unmasking(x, ids_restore, T, temp_embed, mask_token, no_temp_emb, single):
    """amortized_complement_unmasking (multiagent_mae.py:836-867) -> [BA, L, Dd] token rows."""
    bat, k, d = x.shape
    L = ids_restore.shape[1]
    ba = bat // T
    xs = x.reshape(ba, T, k, d)
    if not no_temp_emb:
        xs = xs + temp_embed
    if single:
        cur = xs[:, 0, :L]
        cur = torch.cat([cur, mask_token.expand(ba, L + 1 - cur.shape[1], d)], dim=1)
        return take_rows(cur, ids_restore)
    seq = xs.reshape(ba, T * k, d)[:, :L]
    return take_rows(seq, ids_restore)


def random_unmasking(x, mask, ids_restore, T, temp_embed, mask_token, no_temp_emb, single):
    """amortized_random_unmasking (multiagent_mae.py:752-798).  Returns the CHANNEL-FIRST tensor
    [BA, Dd, L] exactly as the reference does; the caller reinterprets its memory as token rows (Q2)."""
    bat, k, d = x.shape
    L = ids_restore.shape[1]
    ba = bat // T
    full = torch.cat([x, mask_token.expand(bat, L + 1 - k, d)], dim=1)
    full = take_rows(full, ids_restore).reshape(ba, T, L, d)
    if not no_temp_emb:
        full = full + temp_embed
    kept = 1.0 - mask.reshape(ba, T, L)
    cur, filled = full[:, 0], kept[:, 0]
    if not single:
        for t in range(1, T):
            sel = ((filled == 0) & (kept[:, t] > 0)).unsqueeze(-1)
            cur = torch.where(sel, full[:, t], cur)
            filled = filled + kept[:, t]
    return cur.permute(0, 2, 1)


# --------------------------------------------------------------------------------------------
# transformer pieces (timm 0.3.2 definition)
# --------------------------------------------------------------------------------------------
def patch_embed(x, w, b):
    """timm PatchEmbed: Conv2d(k = s = patch) then flatten(2).transpose(1,2)."""
    return F.conv2d(x, w, b, stride=w.shape[-1]).flatten(2).transpose(1, 2)


def attention(x, sd, pre, heads):
    n, L, d = x.shape
    hd = d // heads
    qkv = F.linear(x, sd[pre + "qkv.weight"], sd[pre + "qkv.bias"]).reshape(n, L, 3, heads, hd).permute(2, 0, 3, 1, 4)
    att = ((qkv[0] @ qkv[1].transpose(-2, -1)) * hd ** -0.5).softmax(dim=-1)
    out = (att @ qkv[2]).transpose(1, 2).reshape(n, L, d)
    return F.linear(out, sd[pre + "proj.weight"], sd[pre + "proj.bias"])


def block(x, sd, pre, heads, eps):
    d = x.shape[-1]
    h = F.layer_norm(x, (d,), sd[pre + "norm1.weight"], sd[pre + "norm1.bias"], eps)
    x = x + attention(h, sd, pre + "attn.", heads)
    h = F.layer_norm(x, (d,), sd[pre + "norm2.weight"], sd[pre + "norm2.bias"], eps)
    h = F.gelu(F.linear(h, sd[pre + "mlp.fc1.weight"], sd[pre + "mlp.fc1.bias"]))
    return x + F.linear(h, sd[pre + "mlp.fc2.weight"], sd[pre + "mlp.fc2.bias"])


# --------------------------------------------------------------------------------------------
# the model
# --------------------------------------------------------------------------------------------
def encode(sd, cfg, imgs1, imgs_next, mask_ratio, noise=None, taps=None):
    """forward_encoder_all / forward_encoder_partial (multiagent_mae.py:946-1026)."""
    T = cfg["T"]
    ba, c, hh, ww = imgs1.shape
    x = imgs1.unsqueeze(1)
    if T > 1:
        x = torch.cat([x, imgs_next], dim=1)
    assert x.shape[1] == T                                    # multiagent_mae.py:959
    x = patch_embed(x.reshape(ba * T, c, hh, ww), sd["patch_embed.proj.weight"], sd["patch_embed.proj.bias"])
    if taps is not None:
        taps["tok_patch"] = x
    x = x + sd["pos_embed"][:, 1:]

    def masking(z):
        if cfg["mask"] == "complement":
            return complement_masking(z, T, noise)
        return random_masking(z, mask_ratio, noise)

    if cfg["encode_partial"]:
        x, mask, ids_restore = masking(x)
    for i in range(cfg["depth"]):
        x = block(x, sd, "blocks.%d." % i, cfg["heads"], cfg["eps"])
        if taps is not None:
            taps["enc_blk%d" % i] = x
    x = F.layer_norm(x, (cfg["dim"],), sd["norm.weight"], sd["norm.bias"], cfg["eps"])
    if taps is not None:
        taps["enc_norm"] = x
    if not cfg["encode_partial"]:
        x, mask, ids_restore = masking(x)
    latent = F.relu(F.linear(x, sd["compressor.0.weight"], sd["compressor.0.bias"]))
    return latent, mask, ids_restore


def decode(sd, cfg, latent, mask, ids_restore, taps=None):
    """forward_decoder (multiagent_mae.py:1028-1058) -> logits [BA, 2, C, H, W], class 0 = free."""
    T, p = cfg["T"], cfg["patch"]
    x = F.linear(latent, sd["decompressor.weight"], sd["decompressor.bias"])
    x = F.linear(x, sd["decoder_embed.weight"], sd["decoder_embed.bias"])
    if taps is not None:
        taps["dec_embed"] = x
    args = (T, sd["decoder_temp_embed"], sd["mask_token"], cfg["no_temp_emb"], cfg["decode_singletemp"])
    if cfg["mask"] == "complement":
        x = complement_unmasking(x, ids_restore, *args)
    else:
        x = random_unmasking(x, mask, ids_restore, *args)
    if taps is not None:
        taps["dec_in"] = x
    L = (cfg["img"] // p) ** 2
    x = x.reshape(x.shape[0], L, -1)                          # Q2: plain memory reinterpretation
    x = x + sd["decoder_pos_embed"][:, 1:]
    for i in range(cfg["ddepth"]):
        x = block(x, sd, "decoder_blocks.%d." % i, cfg["dheads"], cfg["eps"])
        if taps is not None:
            taps["dec_blk%d" % i] = x
    x = F.layer_norm(x, (cfg["ddim"],), sd["decoder_norm.weight"], sd["decoder_norm.bias"], cfg["eps"])
    if taps is not None:
        taps["dec_norm"] = x
    occ = unpatchify(F.linear(x, sd["decoder_pred_occ.weight"], sd["decoder_pred_occ.bias"]), p)
    free = unpatchify(F.linear(x, sd["decoder_pred_free.weight"], sd["decoder_pred_free.bias"]), p)
    return torch.stack((free, occ), dim=1)


def vector_quantize(x, codebook, commitment):
    """STARVectorQuantizer.forward (star/models/mae_base.py:122-173), decay = 0 variant (the CLI default,
    completion/train_completion.py:450).  -> (loss, quantized (straight-through), perplexity, indices [M])."""
    flat = x.reshape(-1, codebook.shape[1])
    distances = (torch.sum(flat ** 2, dim=1, keepdim=True) + torch.sum(codebook ** 2, dim=1, keepdim=True).t()
                 - 2 * torch.matmul(flat, codebook.t()))
    indices = torch.argmin(distances, dim=1)
    encodings = torch.zeros(indices.shape[0], codebook.shape[0])
    encodings.scatter_(1, indices.unsqueeze(1), 1)
    quantized = torch.matmul(encodings, codebook).view(x.shape)
    e_latent_loss = F.mse_loss(quantized.detach(), x)
    q_latent_loss = F.mse_loss(quantized, x.detach())
    loss = q_latent_loss + commitment * e_latent_loss
    quantized = x + (quantized - x).detach()
    avg_probs = torch.mean(encodings, dim=0)
    perplexity = torch.exp(-torch.sum(avg_probs * torch.log(avg_probs + 1e-10)))
    return loss, quantized, perplexity, indices


def vector_quantize_ema(x, codebook, ema_w, ema_cluster_size, commitment, decay, training, epsilon=1e-5):
    """STARVectorQuantizerEMA.forward (star/models/mae_base.py:197-253) after its one-off k-means initialisation.
    -> (loss, quantized (straight-through), perplexity, indices, new_codebook, new_ema_w, new_cluster_size); the three
    state tensors are returned unchanged when not training."""
    flat = x.reshape(-1, codebook.shape[1])
    distances = (torch.sum(flat ** 2, dim=1, keepdim=True) + torch.sum(codebook ** 2, dim=1) - 2 * torch.matmul(flat, codebook.t()))
    indices = torch.argmin(distances, dim=1)
    encodings = torch.zeros(indices.shape[0], codebook.shape[0])
    encodings.scatter_(1, indices.unsqueeze(1), 1)
    quantized = torch.matmul(encodings, codebook).view(x.shape)
    new_cb, new_w, new_cs = codebook, ema_w, ema_cluster_size
    if training:
        new_cs = ema_cluster_size * decay + (1 - decay) * torch.sum(encodings, 0)
        n = torch.sum(new_cs)
        new_cs = (new_cs + epsilon) / (n + codebook.shape[0] * epsilon) * n
        dw = torch.matmul(encodings.t(), flat.detach())
        new_w = ema_w * decay + (1 - decay) * dw
        new_cb = new_w / new_cs.unsqueeze(1)
    e_latent_loss = F.mse_loss(quantized.detach(), x)
    loss = commitment * e_latent_loss
    quantized = x + (quantized - x).detach()
    avg_probs = torch.mean(encodings, dim=0)
    perplexity = torch.exp(-torch.sum(avg_probs * torch.log(avg_probs + 1e-10)))
    return loss, quantized, perplexity, indices, new_cb, new_w, new_cs


def run(sd, cfg, imgs1, imgs_next, trans, num_agent, batch, mask_ratio=0.75, infer=False, noise=None, taps=None):
    """forward (multiagent_mae.py:1084-1103) / inference (:1105-1125).
    Returns (loss, result, latent, ind_result, pred).  With cfg["vq"] (VQ-STAR, star/models/VQSTAR.py:421-470) the
    latent is quantised before the decoder, loss = vq + reconstruction, and taps (if given) receives vq_loss,
    perplexity, vq_indices."""
    latent, mask, ids_restore = encode(sd, cfg, imgs1, imgs_next, mask_ratio, noise, taps)
    if cfg.get("vq"):
        if cfg.get("vq_decay", 0.0) > 0.0:     # EMA codebook: the updated state goes to taps["vq_state"]
            vq_loss, quantized, perplexity, vq_idx, ncb, nw, ncs = vector_quantize_ema(
                latent, sd["_vq_star._embedding.weight"].detach(), sd["_vq_star._ema_w"].detach(), sd["_vq_star._ema_cluster_size"],
                cfg["commitment"], cfg["vq_decay"], training=cfg.get("vq_training", not infer))
            if taps is not None:
                taps["vq_state"] = (ncb, nw, ncs)
        else:
            vq_loss, quantized, perplexity, vq_idx = vector_quantize(latent, sd["_vq_star._embedding.weight"], cfg["commitment"])
        pred = decode(sd, cfg, quantized, mask, ids_restore, taps)
        if taps is not None:
            taps["mask"], taps["ids_restore"], taps["pred"] = mask, ids_restore, pred
            taps["vq_loss"], taps["perplexity"], taps["vq_indices"], taps["quantized"] = vq_loss, perplexity, vq_idx, quantized
        loss = vq_loss + F.cross_entropy(pred, imgs1.long())
        ind = torch.argmax(torch.softmax(pred.detach(), dim=1), dim=1)
        fused = late_fusion(ind.numpy().astype(F32), np.asarray(trans), np.asarray(num_agent), batch,
                            ego_maps=imgs1.detach().numpy() if infer else None)
        return loss, torch.from_numpy(fused), latent, ind, pred
    pred = decode(sd, cfg, latent, mask, ids_restore, taps)
    if taps is not None:
        taps["mask"], taps["ids_restore"], taps["pred"] = mask, ids_restore, pred
    loss = F.cross_entropy(pred, imgs1.long())                # Q4: target = own input
    ind = torch.argmax(torch.softmax(pred.detach(), dim=1), dim=1)
    fused = late_fusion(ind.numpy().astype(F32), np.asarray(trans), np.asarray(num_agent), batch,
                        ego_maps=imgs1.detach().numpy() if infer else None)
    return loss, torch.from_numpy(fused), latent, ind, pred


# ---- next row f-1: the joint (latent-level fusion) model -------------------------------------------------------------
def joint_random_unmasking(x, mask, ids_restore, T, temp_embed, mask_token):
    """AmortizedFusionMMAEViT.amortized_random_unmasking (multiagent_mae.py:399-441): as the individual model's, but the
    temporal embedding is always added and the result stays channels-LAST [BA, L, Dd] (no transposed re-read)."""
    return random_unmasking(x, mask, ids_restore, T, temp_embed, mask_token, False, False).permute(0, 2, 1)


def fusion_sum(feat, trans, num_agent, batch):
    """forward_fusion's loop (multiagent_mae.py:497-551) with fusion("sum") (mae_base.py:624-633): feat [A*B, Dd, g, g];
    ego + warped neighbours summed in list order, then >= 0.5 -> 1, < 0.5 -> 0 by masked assignment (no gradient passes);
    slots >= num_agent keep their features (gradient passes)."""
    ab = feat.shape[0]
    A = ab // batch
    out = []
    det = feat.detach().numpy()
    for i in range(A):
        for b in range(batch):
            n = int(num_agent[b, 0])
            if i >= n:
                out.append(feat[i * batch + b])
                continue
            acc = det[i * batch + b].copy()
            for j in range(n):
                if j != i:
                    acc = acc + warp_map(det[j * batch + b], trans[b, i, j])
            fused = acc.copy()
            fused[acc >= 0.5] = 1.0
            fused[acc < 0.5] = 0.0
            out.append(torch.from_numpy(fused.astype(F32)))
    return torch.stack(out)


def conv_pred(x, sd, pre, training, new_stats=None):
    """ConvPred / ConvPred16 (mae_base.py:16-105): (bilinear x2, conv 3x3 pad 1, BatchNorm, ReLU) per stage, no ReLU after
    the last.  new_stats (dict) receives the running statistics a training forward leaves behind."""
    s = 1
    while (pre + "conv%d.weight" % s) in sd:
        last = (pre + "conv%d.weight" % (s + 1)) not in sd
        x = F.interpolate(x, scale_factor=2, mode="bilinear")
        x = F.conv2d(x, sd[pre + "conv%d.weight" % s], sd[pre + "conv%d.bias" % s], padding=1)
        rm, rv = sd[pre + "bn%d.running_mean" % s].detach().clone(), sd[pre + "bn%d.running_var" % s].detach().clone()
        x = F.batch_norm(x, rm, rv, sd[pre + "bn%d.weight" % s], sd[pre + "bn%d.bias" % s], training, 0.1, 1e-5)
        if new_stats is not None:
            new_stats[pre + "bn%d.running_mean" % s], new_stats[pre + "bn%d.running_var" % s] = rm, rv
        if not last:
            x = F.relu(x)
        s += 1
    return x


def run_joint(sd, cfg, imgs1, imgs_next, teacher, trans, num_agent, batch, mask_ratio=0.75, noise=None, training=True, taps=None):
    """AmortizedFusionMMAEViT.forward (multiagent_mae.py:659-681) -> (loss, result int64, mask, pred [BA,2,C,H,W])."""
    T, g = cfg["T"], cfg["img"] // cfg["patch"]
    latent, mask, ids_restore = encode(sd, cfg, imgs1, imgs_next, mask_ratio, noise)
    x = F.linear(latent, sd["decompressor.weight"], sd["decompressor.bias"])
    x = F.linear(x, sd["decoder_embed.weight"], sd["decoder_embed.bias"])
    if cfg["mask"] == "complement":
        x = complement_unmasking(x, ids_restore, T, sd["decoder_temp_embed"], sd["mask_token"], False, False)
    else:
        x = joint_random_unmasking(x, mask, ids_restore, T, sd["decoder_temp_embed"], sd["mask_token"])
    ba, L, dd = x.shape
    if taps is not None:
        taps["feat"] = x
    fused = fusion_sum(x.reshape(ba, g, g, dd).permute(0, 3, 1, 2), np.asarray(trans), np.asarray(num_agent), batch)
    x = fused.permute(0, 2, 3, 1).reshape(ba, L, dd)
    if taps is not None:
        taps["fused"] = x
    x = x + sd["decoder_pos_embed"][:, 1:]
    for i in range(cfg["ddepth"]):
        x = block(x, sd, "decoder_blocks.%d." % i, cfg["dheads"], cfg["eps"])
    x = F.layer_norm(x, (dd,), sd["decoder_norm.weight"], sd["decoder_norm.bias"], cfg["eps"])
    if taps is not None:
        taps["dec_norm"] = x
    x = x.reshape(ba, g, g, dd).permute(0, 3, 1, 2).contiguous()
    stats = {} if taps is not None else None
    occ = conv_pred(x, sd, "decoder_pred_occ.", training, stats)
    free = conv_pred(x, sd, "decoder_pred_free.", training, stats)
    if taps is not None:
        taps["bn_stats"] = stats
    pred = torch.stack((free, occ), dim=1)
    loss = F.cross_entropy(pred, teacher.long())
    result = torch.argmax(torch.softmax(pred.detach(), dim=1), dim=1)
    return loss, result, mask, pred


def trainable(sd):
    """Clone a state dict into leaf tensors; the two sin-cos tables stay frozen (mae_base.py:273,306)."""
    out = {}
    for k, v in sd.items():
        t = torch.as_tensor(v).detach().clone()
        if not t.is_floating_point():          # BatchNorm's num_batches_tracked
            out[k] = t
            continue
        t = t.float()
        t.requires_grad_(k not in ("pos_embed", "decoder_pos_embed") and "running_" not in k)
        out[k] = t
    return out


# ---- in front of the path: dataset densification (NEXT row f-2) -----------------------------------------
def voxel_dense(indices, dims=(256, 256, 13)):
    """One frame exactly as the dataset builds it (star/datasets/MultiTempDet.py:134-143; the same three
    statements at :201-209 for the teacher and :244-247 for past frames): zeros(dims, bool), fancy assignment,
    rot90(k=3), float32.  Parity: these ARE the reference's numpy calls (the dataset class itself cannot be
    imported -- its base class lives in the absent coperception package -- and the reference holds no data
    fixture), so numpy's own semantics are the pin: negative indices wrap, out-of-range raises IndexError."""
    v = np.zeros(dims, dtype=bool)
    indices = np.asarray(indices).reshape(-1, 3)
    v[indices[:, 0], indices[:, 1], indices[:, 2]] = 1
    v = np.rot90(v, 3)
    return v.astype(np.float32)


def past_frames(seq_scenes, idx, time_stamp):
    """MultiTempDet.py:228-233, statement for statement."""
    out = []
    end_frame_idx = - time_stamp + 1 + idx
    current_scene = seq_scenes[idx]
    for nextframe_idx in range(idx - 1, end_frame_idx - 1, -1):
        while nextframe_idx < 0 or seq_scenes[nextframe_idx] != current_scene:
            nextframe_idx = nextframe_idx + 1
        out.append(nextframe_idx)
    return out
