#!/usr/bin/env python3
"""Generate golden vectors by importing the reference STAR model files from /root/reference.

Run ONCE in the build container (the reference never travels to the GPU box):

    python tests/golden/gen_golden.py            # writes tests/golden/*.npz

What this script is: a driver.  It holds NO reference source.  It makes the reference's
`star/models/{mae_base,multiagent_mae}.py`, `star/utils/{pos_embed,lr_sched,metrics}.py` importable
in this container with four small shims:

  1. bare package objects for `star`, `star.models`, `star.utils` whose __path__ points into
     /root/reference/star so that the package __init__ files (which need `coperception`) do not run;
  2. a stand-in `timm.models.vision_transformer` exposing `PatchEmbed` and `Block` -- timm==0.3.2 is a
     third-party dependency that is absent from the container; the classes below restate its public
     0.3.2 definition (pre-LN block, fused qkv Linear with bias, scale hd**-0.5, exact-erf GELU,
     Conv2d(k=s=patch) patch embedding).  The reference has no test that pins this boundary, so the
     goldens pin "reference model files + this declared timm definition" (parity note in DESIGN.md);
  3. `numpy.float = float` (removed in numpy>=1.24, used by pos_embed.py:56);
  4. `torch._six` stub exposing `inf` (removed in torch 2, used by utils/misc.py:21).

Every fixture is data: seeded inputs, the captured mask noise, selected intermediate tensors, outputs.
"""
import importlib
import math
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


# ----------------------------------------------------------------------------------------------
# shim (2): the declared timm-0.3.2 definition of the two classes the reference imports
# ----------------------------------------------------------------------------------------------
class _Mlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.act = nn.GELU()
        self.fc2 = nn.Linear(hidden, dim)

    def forward(self, x):
        return self.fc2(self.act(self.fc1(x)))


class _Attention(nn.Module):
    def __init__(self, dim, num_heads, qkv_bias, qk_scale):
        super().__init__()
        self.num_heads = num_heads
        self.scale = qk_scale or (dim // num_heads) ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)

    def forward(self, x):
        b, n, c = x.shape
        qkv = self.qkv(x).reshape(b, n, 3, self.num_heads, c // self.num_heads).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0], qkv[1], qkv[2]
        attn = ((q @ k.transpose(-2, -1)) * self.scale).softmax(dim=-1)
        return self.proj((attn @ v).transpose(1, 2).reshape(b, n, c))


class Block(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio=4.0, qkv_bias=False, qk_scale=None, drop=0.0, attn_drop=0.0,
                 drop_path=0.0, act_layer=nn.GELU, norm_layer=nn.LayerNorm):
        super().__init__()
        self.norm1 = norm_layer(dim)
        self.attn = _Attention(dim, num_heads, qkv_bias, qk_scale)
        self.norm2 = norm_layer(dim)
        self.mlp = _Mlp(dim, int(dim * mlp_ratio))

    def forward(self, x):
        x = x + self.attn(self.norm1(x))
        return x + self.mlp(self.norm2(x))


class PatchEmbed(nn.Module):
    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768):
        super().__init__()
        self.img_size = (img_size, img_size)
        self.patch_size = (patch_size, patch_size)
        self.num_patches = (img_size // patch_size) ** 2
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)

    def forward(self, x):
        return self.proj(x).flatten(2).transpose(1, 2)


def install_shims():
    for name, sub in (("star", "star"), ("star.models", "star/models"), ("star.utils", "star/utils")):
        m = types.ModuleType(name)
        m.__path__ = [os.path.join(REF, sub)]
        sys.modules[name] = m
    timm = types.ModuleType("timm")
    timm.__version__ = "0.3.2"
    tm = types.ModuleType("timm.models")
    tv = types.ModuleType("timm.models.vision_transformer")
    tv.PatchEmbed, tv.Block = PatchEmbed, Block
    timm.models, tm.vision_transformer = tm, tv
    sys.modules.update({"timm": timm, "timm.models": tm, "timm.models.vision_transformer": tv})
    if not hasattr(np, "float"):
        np.float = float
    six = types.ModuleType("torch._six")
    six.inf = math.inf
    sys.modules["torch._six"] = six


# ----------------------------------------------------------------------------------------------
# synthetic inputs (same law as bench.py / SURVEY 8d, at reduced size)
# ----------------------------------------------------------------------------------------------
def rigid(yaw, tx, ty):
    c, s = math.cos(yaw), math.sin(yaw)
    return np.array([[c, -s, 0, tx], [s, c, 0, ty], [0, 0, 1, 0], [0, 0, 0, 1]], dtype=np.float64)


def make_trans(rng, B, A, special=True):
    t = np.zeros((B, A, A, 4, 4), dtype=np.float64)
    for b in range(B):
        for i in range(A):
            for j in range(A):
                if i == j:
                    t[b, i, j] = np.eye(4)
                else:
                    t[b, i, j] = rigid(rng.uniform(-math.pi, math.pi), rng.uniform(-20, 20), rng.uniform(-20, 20))
    if special and A >= 3:
        t[0, 0, 1] = rigid(0.0, 8.0, -12.0)           # pure translation
        t[0, 0, 2] = rigid(math.pi / 2, 0.0, 0.0)     # pure rotation by 90 degrees
        t[0, 1, 0] = rigid(0.3, 0.0, 0.0)             # pure rotation
        t[0, 1, 2] = rigid(0.0, 16.0, 16.0)           # translation of exactly half a cell at 64x64: .5 ties
    return t


def make_inputs(seed, B, A, T, S, occ=0.10):
    g = torch.Generator().manual_seed(seed)
    bev = (torch.rand(A * B, 13, S, S, generator=g) < occ).float()
    nxt = (torch.rand(A * B, max(T - 1, 0), 13, S, S, generator=g) < occ).float() if T > 1 else torch.zeros(A * B, 0)
    teacher = (torch.rand(A * B, 13, S, S, generator=g) < 2 * occ).float()
    return bev, nxt, teacher


def pack(t):
    a = t.detach().cpu().numpy()
    return np.packbits(a.astype(np.uint8).reshape(-1)), np.array(a.shape, dtype=np.int64)


# ----------------------------------------------------------------------------------------------
_SD_DONE = set()


def gen_model_case(mm, name, ctor, T, mask, flags, B, A, num_agent, S, mask_ratio, mode, seed, with_grads, model=None, sd_name=None,
                   inputs=None):
    """model / sd_name / inputs: a prepared (e.g. trained) reference model and its input maps instead of the seeded
    random-init model and random-bit maps (gen_golden_trained.py)."""
    if model is None:
        sd_name = "sd_%s_T%d" % (ctor.__name__, T)
        torch.manual_seed(0)
        model = ctor(img_size=S, time_stamp=T, mask_method=mask, **flags)
        # make every parameter non-trivial so that unused/zero-init ones are exercised too
        torch.manual_seed(7)
        with torch.no_grad():
            model.decoder_temp_embed.normal_(std=0.02)
            for n_, p in model.named_parameters():
                if n_.endswith(".bias") or "norm" in n_:
                    p.add_(torch.randn_like(p) * 0.05)
    model.train() if mode == "train" else model.eval()
    rng = np.random.RandomState(seed)
    bev, nxt, teacher = make_inputs(seed, B, A, T, S) if inputs is None else inputs
    trans = torch.from_numpy(make_trans(rng, B, A))
    na = torch.tensor(num_agent, dtype=torch.int64).reshape(B, 1).repeat(1, A)

    taps = {}
    hooks = []

    def tap(key):
        def f(_m, _i, o):
            taps[key] = o.detach().clone()
        return f

    hooks.append(model.patch_embed.register_forward_hook(tap("tok_patch")))
    for i, blk in enumerate(model.blocks):
        hooks.append(blk.register_forward_hook(tap("enc_blk%d" % i)))
    hooks.append(model.norm.register_forward_hook(tap("enc_norm")))
    hooks.append(model.decoder_embed.register_forward_hook(tap("dec_embed")))
    for i, blk in enumerate(model.decoder_blocks):
        hooks.append(blk.register_forward_hook(tap("dec_blk%d" % i)))
    hooks.append(model.decoder_norm.register_forward_hook(tap("dec_norm")))

    # capture the unmasked decoder input and the logits
    orig_unmask, orig_dec = model.unmasking_handle, model.forward_decoder

    def unmask(x, m, ids):
        r = orig_unmask(x, m, ids)
        taps["dec_in"] = r.detach().clone()
        taps["mask"] = m.detach().clone()
        taps["ids_restore"] = ids.detach().clone()
        return r

    def dec(latent, m, ids):
        r = orig_dec(latent, m, ids)
        taps["pred"] = r.detach().clone()
        return r

    model.unmasking_handle, model.forward_decoder = unmask, dec

    noises = []
    orig_rand = torch.rand

    def rec_rand(*a, **k):
        r = orig_rand(*a, **k)
        noises.append(r.clone())
        return r

    torch.manual_seed(123)
    torch.rand = rec_rand
    try:
        if mode == "train":
            loss, result, latent, ind = model(bev, nxt, teacher, trans, na, B, mask_ratio=mask_ratio)
            loss.backward()
        else:
            with torch.no_grad():
                loss, result, latent, ind = model.inference(bev, nxt, teacher, trans, na, B, mask_ratio=mask_ratio)
    finally:
        torch.rand = orig_rand
    for h in hooks:
        h.remove()

    out = {}
    out["bev_bits"], out["bev_shape"] = pack(bev)
    if T > 1:
        out["next_bits"], out["next_shape"] = pack(nxt)
    out["teacher_bits"], out["teacher_shape"] = pack(teacher)
    out["trans"] = trans.numpy()
    out["num_agent"] = na.numpy()
    out["noise"] = np.stack([n.numpy() for n in noises]) if noises else np.zeros((0,), np.float32)
    out["meta"] = np.array([B, A, T, S, int(mask == "random"), int(mode == "train")], dtype=np.int64)
    out["mask_ratio"] = np.array(mask_ratio, dtype=np.float64)
    out["flags"] = np.array([int(flags.get(k, False)) for k in ("encode_partial", "no_temp_emb", "decode_singletemp")])
    out["loss"] = np.array(loss.item(), dtype=np.float64)
    out["latent"] = latent.detach().numpy()
    out["ind_bits"], out["ind_shape"] = pack(ind)
    out["result_bits"], out["result_shape"] = pack(result)
    for k, v in taps.items():
        out["tap_" + k] = v.numpy()
    sd_path = os.path.join(OUT, sd_name + ".npz")   # one state dict per (constructor, T): flags do not change it
    if not os.path.exists(sd_path) or sd_name not in _SD_DONE:
        np.savez_compressed(sd_path, **{k: v.numpy() for k, v in model.state_dict().items()})
        _SD_DONE.add(sd_name)
    out["sd_name"] = np.array(sd_name)
    if with_grads:
        for k, p in model.named_parameters():
            out["grad_" + k] = p.grad.numpy() if p.grad is not None else np.zeros((0,), np.float32)
    # occupancy IoU of the fused result against the teacher, through the reference evaluator
    metrics = importlib.import_module("star.utils.metrics")
    M = metrics.Metrics(2, 1)
    M.add_batch(result.detach().clone(), teacher.clone())
    M.update_IoU()
    out["iou"] = np.asarray(M.get_average_IoU(), dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print("wrote", name, "loss", float(out["loss"]), "iou", out["iou"])


def gen_warp_case(mm, seed=11):
    """Full-resolution (13x256x256) pose-warp + union fusion, through the reference's own late_fusion /
    ego_late_fusion (multiagent_mae.py:870-944) on integer maps; bit-packed."""
    torch.manual_seed(0)
    model = mm.amortized_ind_patch8_light(time_stamp=1, mask_method="complement")
    B, A, S = 2, 5, 256
    g = torch.Generator().manual_seed(seed)
    ind = (torch.rand(A * B, 13, S, S, generator=g) < 0.03).long()
    gt = (torch.rand(A * B, 13, S, S, generator=g) < 0.03).float()
    rng = np.random.RandomState(seed)
    trans = make_trans(rng, B, A)
    trans[0, 1, 2] = rigid(0.0, 4.0, 4.0)   # exactly half a cell at 256x256: .5 ties in both axes
    trans[1, 0, 1] = rigid(math.pi, 0.0, 0.0)
    trans[1, 2, 3] = rigid(math.pi / 4, 32.0, -32.0)
    trans = torch.from_numpy(trans)
    na = torch.tensor([[5] * A, [3] * A], dtype=torch.int64)
    late = model.late_fusion(ind, trans, na, B)
    ego = model.ego_late_fusion(ind, gt, trans, na, B)
    # single warps straight through feature_transformation (mae_base.py:589-614)
    x = ind.float().reshape(A, B, 13, S, S).transpose(0, 1).contiguous()   # [B, A, ...]
    single = []
    pairs = [(0, 0, 1), (0, 0, 2), (0, 1, 0), (0, 1, 2), (1, 0, 1), (1, 2, 3), (1, 4, 0)]
    for b, i, j in pairs:
        single.append(model.feature_transformation(b, j, x, trans[b, i], torch.device("cpu"), (1, 13, S, S)))
    out = {"trans": trans.numpy(), "num_agent": na.numpy(), "pairs": np.array(pairs, dtype=np.int64)}
    out["ind_bits"], out["shape"] = pack(ind)
    out["gt_bits"], _ = pack(gt)
    out["late_bits"], _ = pack(late)
    out["ego_bits"], _ = pack(ego)
    out["single_bits"], out["single_shape"] = pack(torch.stack(single))
    # a float32-typed trans matrix goes through a different rounding path (theta built in source dtype)
    late32 = model.late_fusion(ind, trans.float(), na, B)
    out["late32_bits"], _ = pack(late32)
    np.savez_compressed(os.path.join(OUT, "warp_full256.npz"), **out)
    print("wrote warp_full256", float(late.sum()), float(ego.sum()), float(late32.sum()))


def gen_standalone(mm):
    pe = importlib.import_module("star.utils.pos_embed")
    ls = importlib.import_module("star.utils.lr_sched")
    metrics = importlib.import_module("star.utils.metrics")
    out = {}
    # stored as the float32 the model holds (mae_base.py:348 casts with .float())
    for d in (384, 256, 32):
        out["pos_%d_32" % d] = pe.get_2d_sincos_pos_embed(d, 32, cls_token=True).astype(np.float32)
    out["pos_768_16"] = pe.get_2d_sincos_pos_embed(768, 16, cls_token=True).astype(np.float32)
    out["pos_48_4"] = pe.get_2d_sincos_pos_embed(48, 4, cls_token=True).astype(np.float32)
    # lr schedule table
    args = types.SimpleNamespace(lr=1e-3, min_lr=1e-5, warmup_epochs=4, nepoch=30)

    class _Opt:
        param_groups = [{"lr": 0.0}, {"lr": 0.0, "lr_scale": 0.5}]

    eps = np.linspace(0.0, 30.0, 241)
    out["lr_epochs"] = eps
    out["lr_values"] = np.array([ls.adjust_learning_rate(_Opt, float(e), args) for e in eps], dtype=np.float64)
    # patchify / unpatchify
    torch.manual_seed(0)
    model = mm.amortized_ind_patch8_light(time_stamp=1, mask_method="complement")
    x = torch.arange(2 * 13 * 256 * 256, dtype=torch.float32).reshape(2, 13, 256, 256) % 251
    p = model.patchify(x)
    out["patchify_probe"] = p[:, ::97, ::53].numpy()
    assert torch.equal(model.unpatchify(p), x)
    # occupancy IoU, 3 scales, two batches
    g = torch.Generator().manual_seed(5)
    M = metrics.Metrics(2, 1)
    preds, tgts = [], []
    for _ in range(2):
        pr = (torch.rand(3, 13, 256, 256, generator=g) < 0.05).float()
        tg = (torch.rand(3, 13, 256, 256, generator=g) < 0.05).float()
        tg = torch.maximum(tg, pr * (torch.rand(3, 13, 256, 256, generator=g) < 0.6).float())
        preds.append(pr)
        tgts.append(tg)
        M.add_batch(pr.clone(), tg.clone())
        M.update_IoU()
    out["iou_pred_bits"], out["iou_shape"] = pack(torch.stack(preds))
    out["iou_tgt_bits"], _ = pack(torch.stack(tgts))
    out["iou_every_batch"] = np.asarray(M.every_batch_IoU, dtype=np.float64)
    out["iou_avg"] = np.asarray(M.get_average_IoU(), dtype=np.float64)
    # model inventory: parameter names / shapes of the constructors on the hot path
    names = []
    for fn in ("amortized_ind_patch8_light", "amortized_ind_patch8_shallow", "amortized_ind_patch16_light"):
        m = getattr(mm, fn)(time_stamp=2, mask_method="complement")
        names.append(fn + "|" + ";".join("%s:%s" % (k, "x".join(map(str, v.shape))) for k, v in m.state_dict().items()))
        names.append(fn + "|nparams=%d|trainable=%d" % (sum(p.numel() for p in m.parameters()),
                                                         sum(p.numel() for p in m.parameters() if p.requires_grad)))
    out["inventory"] = np.array(names)
    np.savez_compressed(os.path.join(OUT, "standalone.npz"), **out)
    print("wrote standalone")


def main():
    install_shims()
    mm = importlib.import_module("star.models.multiagent_mae")
    from functools import partial

    def tiny(**kw):   # L = (S/8)^2 tokens; hd = 16 in both halves
        return mm.AmortizedIndivMMAEViT(patch_size=8, in_chans=13, embed_dim=32, depth=2, num_heads=2,
                                        decoder_embed_dim=32, decoder_depth=2, decoder_num_heads=2, mlp_ratio=4,
                                        norm_layer=partial(nn.LayerNorm, eps=1e-6), decoder_head="mlp", **kw)

    def small(**kw):  # hd = 32 encoder, 16 decoder
        return mm.AmortizedIndivMMAEViT(patch_size=8, in_chans=13, embed_dim=64, depth=2, num_heads=2,
                                        decoder_embed_dim=48, decoder_depth=1, decoder_num_heads=3, mlp_ratio=4,
                                        norm_layer=partial(nn.LayerNorm, eps=1e-6), decoder_head="mlp", **kw)

    gen_standalone(mm)
    gen_warp_case(mm)
    C = dict(mask_ratio=0.75, A=5)
    gen_model_case(mm, "tiny_T1_comp_train", tiny, 1, "complement", {}, B=1, num_agent=[5], S=32, mode="train", seed=1, with_grads=True, **C)
    gen_model_case(mm, "tiny_T1_comp_infer", tiny, 1, "complement", {}, B=2, num_agent=[5, 3], S=32, mode="infer", seed=2, with_grads=False, **C)
    gen_model_case(mm, "tiny_T2_comp_train", tiny, 2, "complement", {}, B=2, num_agent=[3, 5], S=32, mode="train", seed=3, with_grads=True, **C)
    gen_model_case(mm, "tiny_T3_comp_train", tiny, 3, "complement", {}, B=1, num_agent=[5], S=32, mode="train", seed=4, with_grads=True, **C)
    gen_model_case(mm, "tiny_T3_comp_infer", tiny, 3, "complement", {}, B=2, num_agent=[4, 5], S=32, mode="infer", seed=5, with_grads=False, **C)
    gen_model_case(mm, "tiny_T3_rand_train", tiny, 3, "random", {}, B=1, num_agent=[5], S=32, mode="train", seed=6, with_grads=True, **C)
    gen_model_case(mm, "tiny_T3_rand_infer", tiny, 3, "random", {}, B=2, num_agent=[5, 2], S=32, mode="infer", seed=7, with_grads=False, **C)
    gen_model_case(mm, "tiny_T3_partial_train", tiny, 3, "complement", {"encode_partial": True}, B=1, num_agent=[5], S=32, mode="train", seed=8, with_grads=True, **C)
    gen_model_case(mm, "tiny_T3_single_infer", tiny, 3, "complement", {"decode_singletemp": True}, B=1, num_agent=[5], S=32, mode="infer", seed=9, with_grads=False, **C)
    gen_model_case(mm, "tiny_T2_notemp_train", tiny, 2, "complement", {"no_temp_emb": True}, B=1, num_agent=[5], S=32, mode="train", seed=10, with_grads=True, **C)
    gen_model_case(mm, "tiny_T2_rand_partial_train", tiny, 2, "random", {"encode_partial": True}, B=1, num_agent=[5], S=32, mode="train", seed=12, with_grads=True, mask_ratio=0.5, A=5)
    gen_model_case(mm, "small_T2_comp_train", small, 2, "complement", {}, B=1, num_agent=[3], S=64, mode="train", seed=13, with_grads=True, mask_ratio=0.75, A=3)
    gen_model_case(mm, "small_T1_comp_infer", small, 1, "complement", {}, B=1, num_agent=[2], S=64, mode="infer", seed=14, with_grads=False, mask_ratio=0.75, A=3)


if __name__ == "__main__":
    main()
