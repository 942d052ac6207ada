#!/usr/bin/env python3
"""Golden vectors from TRAINED reference models (run once in the build container, after gen_golden.py):

    python tests/golden/gen_golden_trained.py        # writes tests/golden/trained_*.npz + sd_trained_*.npz

The random-init fixtures of gen_golden.py have logit margins clustered around zero, where a bf16 rounding flips a decision;
a trained model is confident almost everywhere.  This driver imports the reference's model files from /root/reference with
the shims of gen_golden.py, trains the reference `tiny` / `small` models on the CPU for a few hundred AdamW steps
(betas (0.9, 0.95), timm's weight-decay split, as completion/train_completion.py:124-125) on the learnable structured law of
star_amd.synthetic.structured_bev, then captures the same taps / outputs / gradients as the other fixtures on held-out maps.
It holds no reference source; the fixtures are data.
"""
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
from star_amd.synthetic import rigid_poses, structured_bev  # noqa: E402


def add_weight_decay(model, wd):   # timm 0.3.2 optim_factory.add_weight_decay (declared definition, see gen_golden.py shim note)
    decay, no_decay = [], []
    for n, p in model.named_parameters():
        if p.requires_grad:
            (no_decay if p.dim() == 1 or n.endswith(".bias") else decay).append(p)
    return [{"params": no_decay, "weight_decay": 0.0}, {"params": decay, "weight_decay": wd}]


def inputs(seed, B, A, T, S):
    ab = A * B
    fr = structured_bev(ab * (T + 1), S, 13, seed)
    bev = fr[:ab].contiguous()
    nxt = fr[ab:ab * T].reshape(ab, T - 1, 13, S, S).contiguous() if T > 1 else torch.zeros(ab, 0)
    teacher = torch.maximum(bev, fr[ab * T:])
    return bev, nxt, teacher


def train(model, T, S, steps, B, A, mask_ratio, seed):
    opt = torch.optim.AdamW(add_weight_decay(model, 0.05), lr=2e-3, betas=(0.9, 0.95))
    model.train()
    na = torch.full((B, A), A, dtype=torch.int64)
    for it in range(steps):
        bev, nxt, teacher = inputs(seed * 100000 + it, B, A, T, S)
        trans = rigid_poses(B, A, it)
        for g in opt.param_groups:
            g["lr"] = 2e-3 * min(1.0, (it + 1) / 20.0) * (0.5 * (1 + np.cos(np.pi * it / steps)) * 0.9 + 0.1)
        loss = model(bev, nxt, teacher, trans, na, B, mask_ratio=mask_ratio)[0]
        opt.zero_grad()
        loss.backward()
        opt.step()
        if it % 50 == 0 or it == steps - 1:
            print("  step %4d loss %.4f" % (it, loss.item()), flush=True)
    return model


def main():
    G.install_shims()
    mm = importlib.import_module("star.models.multiagent_mae")

    def tiny(**kw):
        return mm.AmortizedIndivMMAEViT(patch_size=8, in_chans=13, embed_dim=32, depth=2, num_heads=2, decoder_embed_dim=32,
                                        decoder_depth=2, decoder_num_heads=2, mlp_ratio=4, norm_layer=partial(nn.LayerNorm, eps=1e-6),
                                        decoder_head="mlp", **kw)

    def small(**kw):
        return mm.AmortizedIndivMMAEViT(patch_size=8, in_chans=13, embed_dim=64, depth=2, num_heads=2, decoder_embed_dim=48,
                                        decoder_depth=1, decoder_num_heads=3, mlp_ratio=4, norm_layer=partial(nn.LayerNorm, eps=1e-6),
                                        decoder_head="mlp", **kw)

    # (fixture tag, ctor, T, mask, mask_ratio, S, training steps, [(case suffix, mode, B, A, num_agent, seed)])
    plans = [
        ("small_T2_comp", small, 2, "complement", 0.75, 64, 400, [("train", "train", 1, 3, [3], 901), ("infer", "infer", 2, 3, [3, 2], 902)]),
        ("small_T1_comp", small, 1, "complement", 0.75, 64, 400, [("infer", "infer", 2, 3, [3, 3], 903), ("train", "train", 1, 3, [3], 904)]),
        ("tiny_T3_rand", tiny, 3, "random", 0.75, 32, 400, [("infer", "infer", 2, 5, [5, 4], 905), ("train", "train", 1, 5, [5], 906)]),
        ("tiny_T3_rand90", tiny, 3, "random", 0.9, 32, 400, [("infer", "infer", 2, 5, [5, 5], 907)]),
    ]
    for tag, ctor, T, mask, ratio, S, steps, cases in plans:
        torch.manual_seed(0)
        model = ctor(img_size=S, time_stamp=T, mask_method=mask)
        print("training", tag, flush=True)
        train(model, T, S, steps, B=4, A=3, mask_ratio=ratio, seed=len(tag))
        sd_name = "sd_trained_%s" % tag
        np.savez_compressed(os.path.join(G.OUT, sd_name + ".npz"), **{k: v.numpy() for k, v in model.state_dict().items()})
        G._SD_DONE.add(sd_name)
        for suffix, mode, B, A, num_agent, seed in cases:
            model.zero_grad(set_to_none=True)
            G.gen_model_case(mm, "trained_%s_%s" % (tag, suffix), ctor, T, mask, {}, B=B, A=A, num_agent=num_agent, S=S, mask_ratio=ratio,
                             mode=mode, seed=seed, with_grads=mode == "train", model=model, sd_name=sd_name, inputs=inputs(seed, B, A, T, S))


if __name__ == "__main__":
    main()
