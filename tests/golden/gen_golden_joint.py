#!/usr/bin/env python3
"""Golden vectors of the JOINT model (AmortizedFusionMMAEViT + ConvPred / ConvPred16), captured by importing the reference's
model files from /root/reference with the shims of gen_golden.py.  Run once in the build container:

    python tests/golden/gen_golden_joint.py          # writes tests/golden/joint*.npz + sd_joint*.npz

The reference hard-codes 512 input channels for the conv head; to keep the fixtures small the model is built with a
64-channel decoder and its two heads are replaced by the reference's OWN ConvPred / ConvPred16 classes constructed with
input_chans=64 (same code, other argument).  BatchNorm runs on batch statistics (train) or on non-trivial running
statistics (eval).  This driver holds no reference source; the fixtures are data."""
import importlib
import os
import sys
from functools import partial

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import gen_golden as G  # noqa: E402
from star_amd.synthetic import structured_bev  # noqa: E402


def build(mm, mb, patch, S, T, mask):
    torch.manual_seed(0)
    model = mm.AmortizedFusionMMAEViT(img_size=S, patch_size=patch, in_chans=13, embed_dim=32, depth=1, num_heads=2, decoder_embed_dim=64,
                                      decoder_depth=1, decoder_num_heads=2, mlp_ratio=4, norm_layer=partial(nn.LayerNorm, eps=1e-6),
                                      decoder_head="conv3", time_stamp=T, mask_method=mask)
    head = mb.ConvPred if patch == 8 else mb.ConvPred16
    torch.manual_seed(1)
    model.decoder_pred_occ = head(input_chans=64)
    model.decoder_pred_free = head(input_chans=64)
    torch.manual_seed(7)
    with torch.no_grad():
        model.decoder_temp_embed.normal_(std=0.3)
        # features spread around the 0.5 threshold so that the thresholded latent is not all zeros
        model.decoder_embed.bias.add_(0.25)
        model.decoder_embed.weight.mul_(3.0)
        for n_, p in model.named_parameters():
            if n_.endswith(".bias") or "norm" in n_ or ".bn" in n_:
                p.add_(torch.randn_like(p) * 0.05)
        for n_, b in model.named_buffers():
            if n_.endswith("running_mean"):
                b.normal_(std=0.2)
            elif n_.endswith("running_var"):
                b.uniform_(0.5, 1.5)
    return model


def case(mm, mb, name, patch, S, T, mask, mask_ratio, B, A, num_agent, train, seed):
    model = build(mm, mb, patch, S, T, mask)
    model.train() if train else model.eval()
    sd0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
    ab = A * B
    fr = structured_bev(ab * (T + 1), S, 13, seed, walls=(2, 5), boxes=(2, 6), noise=0.02)
    bev = fr[:ab].contiguous()
    nxt = fr[ab:ab * T].reshape(ab, T - 1, 13, S, S).contiguous() if T > 1 else torch.zeros(ab, 0)
    teacher = torch.maximum(bev, fr[ab * T:])
    trans = torch.from_numpy(G.make_trans(np.random.RandomState(seed), B, A, special=False))
    na = torch.tensor(num_agent, dtype=torch.int64).reshape(B, 1).repeat(1, A)
    taps, noises = {}, []
    orig_unmask, orig_fusion = model.unmasking_handle, model.forward_fusion

    def unmask(x, m, ids):
        r = orig_unmask(x, m, ids)
        taps["feat"], taps["ids_restore"] = r.detach().clone(), ids.detach().clone()
        return r

    def fusion(*a):
        r = orig_fusion(*a)
        taps["fused"] = r.detach().clone()
        return r

    model.unmasking_handle, model.forward_fusion = unmask, fusion
    h = model.decoder_norm.register_forward_hook(lambda _m, _i, o: taps.__setitem__("dec_norm", o.detach().clone()))
    orig_rand = torch.rand

    def rec_rand(*a, **k):
        r = orig_rand(*a, **k)
        noises.append(r.clone())
        return r

    torch.manual_seed(123)
    torch.rand = rec_rand
    try:
        with torch.set_grad_enabled(train):
            loss, result, mask_out, pred = model(bev, nxt, teacher, trans, na, B, mask_ratio=mask_ratio)
        if train:
            loss.backward()
    finally:
        torch.rand = orig_rand
    h.remove()
    out = {"meta": np.array([B, A, T, S, patch, int(mask == "random"), int(train)], dtype=np.int64), "mask_ratio": np.array(mask_ratio),
           "trans": trans.numpy(), "num_agent": na.numpy(), "noise": noises[0].numpy(), "loss": np.array(loss.item(), dtype=np.float64),
           "result": result.numpy().astype(np.uint8), "mask": mask_out.numpy(), "pred": pred.detach().numpy()}
    out["bev_bits"], out["bev_shape"] = G.pack(bev)
    if T > 1:
        out["next_bits"], out["next_shape"] = G.pack(nxt)
    out["teacher_bits"], out["teacher_shape"] = G.pack(teacher)
    for k, v in taps.items():
        out["tap_" + k] = v.numpy()
    for k, v in sd0.items():
        out["sd_" + k] = v.numpy()
    for k, v in model.state_dict().items():          # what a training forward leaves in the BatchNorm buffers
        if "running_" in k or "num_batches" in k:
            out["after_" + k] = v.numpy()
    if train:
        for k, p in model.named_parameters():
            out["grad_" + k] = p.grad.numpy() if p.grad is not None else np.zeros((0,), np.float32)
    np.savez_compressed(os.path.join(G.OUT, name + ".npz"), **out)
    print("wrote", name, "loss", loss.item(), "fused ones", float(taps["fused"].mean()), "occupied decisions", float(result.float().mean()))


def main():
    G.install_shims()
    mm = importlib.import_module("star.models.multiagent_mae")
    mb = importlib.import_module("star.models.mae_base")
    case(mm, mb, "joint_T1_comp_train", 8, 32, 1, "complement", 0.75, B=2, A=3, num_agent=[3, 2], train=True, seed=31)
    case(mm, mb, "joint_T2_rand_train", 8, 32, 2, "random", 0.5, B=2, A=3, num_agent=[2, 3], train=True, seed=32)
    case(mm, mb, "joint_T2_comp_eval", 8, 32, 2, "complement", 0.75, B=1, A=3, num_agent=[3], train=False, seed=33)
    case(mm, mb, "joint16_T3_comp_train", 16, 64, 3, "complement", 0.75, B=1, A=3, num_agent=[2], train=True, seed=34)


if __name__ == "__main__":
    main()
