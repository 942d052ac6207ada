#!/usr/bin/env python3
"""Golden vectors for the VQ-STAR variant, from the reference's own star/models/VQSTAR.py and
STARVectorQuantizer (star/models/mae_base.py:107-173) imported from /root/reference with the shims of
gen_golden.py (same rules: a driver, no reference source; run once in the build container).

    python tests/golden/gen_golden_vq.py        # writes tests/golden/vq_*.npz
"""
import importlib
import os
import sys
from functools import partial

import numpy as np
import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_golden as G  # noqa: E402


def gen_quantizer(mb):
    """The quantiser alone: forward outputs and both gradients, K not a multiple of the chunk, ties included."""
    out = {}
    for name, M, E, K, cc, seed in (("a", 300, 32, 200, 0.25, 0), ("b", 517, 64, 96, 0.5, 1), ("c", 64, 8, 16, 0.25, 2)):
        torch.manual_seed(seed)
        vq = mb.STARVectorQuantizer(K, E, cc)
        x = torch.relu(torch.randn(3, M // 3 if M % 3 == 0 else 1, E)) if M % 3 == 0 else torch.relu(torch.randn(M, E))
        x = x.reshape(-1, E).clone()
        with torch.no_grad():
            vq._embedding.weight[5].copy_(vq._embedding.weight[3])       # duplicate code: argmin must take the first
            x[7].copy_(vq._embedding.weight[3].clamp_min(0))
        x.requires_grad_(True)
        vq.train()
        loss, quantized, perplexity, enc = vq(x)
        w = torch.randn_like(quantized)
        (loss + (quantized * w).sum()).backward()
        out.update({name + "_x": x.detach().numpy(), name + "_codebook": vq._embedding.weight.detach().numpy(), name + "_w": w.numpy(),
                    name + "_cc": np.array(cc), name + "_loss": np.array(loss.item()), name + "_perplexity": np.array(perplexity.item()),
                    name + "_idx": enc.argmax(1).numpy(), name + "_quantized": quantized.detach().numpy(),
                    name + "_dx": x.grad.numpy(), name + "_dcodebook": vq._embedding.weight.grad.numpy()})
        d = (x.detach() ** 2).sum(1, keepdim=True) + (vq._embedding.weight.detach() ** 2).sum(1, keepdim=True).t() - 2 * x.detach() @ vq._embedding.weight.detach().t()
        top2 = torch.topk(d, 2, dim=1, largest=False).values
        out[name + "_margin"] = (top2[:, 1] - top2[:, 0]).numpy()
    np.savez_compressed(os.path.join(G.OUT, "vq_quantizer.npz"), **out)
    print("wrote vq_quantizer")


def gen_model_case(VQ, name, T, mask, B, A, num_agent, S, mode, seed, K, E, cc, mask_ratio=0.75, decay=0.):
    torch.manual_seed(0)
    model = VQ.VQSTARViT(img_size=S, patch_size=8, in_chans=13, embed_dim=32, depth=2, num_heads=2, decoder_embed_dim=32, decoder_depth=2,
                         decoder_num_heads=2, mlp_ratio=4, norm_layer=partial(nn.LayerNorm, eps=1e-6), decoder_head="mlp", time_stamp=T,
                         mask_method=mask, decay=decay, commitment_cost=cc, num_vq_embeddings=K, vq_embedding_dim=E)
    torch.manual_seed(7)
    with torch.no_grad():
        model.decoder_temp_embed.normal_(std=0.02)
        for n_, p in model.named_parameters():
            if n_.endswith(".bias") or "norm" in n_:
                p.add_(torch.randn_like(p) * 0.05)
        model._vq_star._embedding.weight.mul_(0.3).abs_()       # codes inside the ReLU latent's range: many codes in use
        if decay > 0:      # EMA variant: skip the k-means re-initialisation (scipy RNG), start from a consistent state
            model._vq_star.data_initialized = 1
            model._vq_star._ema_cluster_size.fill_(1.5)
            model._vq_star._ema_w.data.copy_(model._vq_star._embedding.weight * 1.5)
    model.train() if mode == "train" else model.eval()
    sd_before = {k: v.detach().clone() for k, v in model.state_dict().items()}
    rng = np.random.RandomState(seed)
    bev, nxt, teacher = G.make_inputs(seed, B, A, T, S)
    trans = torch.from_numpy(G.make_trans(rng, B, A))
    na = torch.tensor(num_agent, dtype=torch.int64).reshape(B, 1).repeat(1, A)
    taps = {}
    orig_vq = model._vq_star.forward

    def vq_fwd(latent):
        r = orig_vq(latent)
        taps["latent"] = latent.detach().clone()
        taps["vq_loss"], taps["quantized"], taps["perplexity"] = r[0].detach().clone(), r[1].detach().clone(), r[2].detach().clone()
        taps["vq_indices"] = r[3].argmax(1).clone()
        return r

    model._vq_star.forward = vq_fwd
    orig_dec = model.forward_decoder

    def dec(latent, m, ids):
        r = orig_dec(latent, m, ids)
        taps["pred"], taps["mask"], taps["ids_restore"] = r.detach().clone(), m.detach().clone(), ids.detach().clone()
        return r

    model.forward_decoder = dec
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
            loss, result, mask_out, ind, perplexity = model(bev, nxt, teacher, trans, na, B, mask_ratio=mask_ratio)
            loss.backward()
            encodings = None
        else:
            with torch.no_grad():
                loss, result, ind, perplexity, encodings = model.inference(bev, nxt, teacher, trans, na, B, mask_ratio=mask_ratio)
    finally:
        torch.rand = orig_rand
    out = {}
    out["bev_bits"], out["bev_shape"] = G.pack(bev)
    if T > 1:
        out["next_bits"], out["next_shape"] = G.pack(nxt)
    out["teacher_bits"], out["teacher_shape"] = G.pack(teacher)
    out["trans"], out["num_agent"] = trans.numpy(), na.numpy()
    out["noise"] = np.stack([n.numpy() for n in noises])
    out["meta"] = np.array([B, A, T, S, int(mask == "random"), int(mode == "train"), K, E], dtype=np.int64)
    out["mask_ratio"], out["cc"] = np.array(mask_ratio, dtype=np.float64), np.array(cc, dtype=np.float64)
    out["loss"], out["perplexity"] = np.array(loss.item(), dtype=np.float64), np.array(perplexity.item(), dtype=np.float64)
    out["ind_bits"], out["ind_shape"] = G.pack(ind)
    out["result_bits"], out["result_shape"] = G.pack(result)
    if encodings is not None:
        assert torch.equal(encodings.argmax(1), taps["vq_indices"]) and float(encodings.sum()) == encodings.shape[0]
        out["encodings_shape"] = np.array(encodings.shape)
    for k, v in taps.items():
        out["tap_" + k] = v.numpy()
    lat = taps["latent"].reshape(-1, E)
    cb = model._vq_star._embedding.weight.detach()
    d = (lat ** 2).sum(1, keepdim=True) + (cb ** 2).sum(1, keepdim=True).t() - 2 * lat @ cb.t()
    top2 = torch.topk(d, 2, dim=1, largest=False).values
    out["vq_margin"] = (top2[:, 1] - top2[:, 0]).numpy()
    if decay > 0:      # state AFTER the step (mae_base.py:228-241 replaces the three tensors in training mode)
        out["decay"] = np.array(decay)
        out["post_embedding"] = model._vq_star._embedding.weight.detach().numpy()
        out["post_ema_w"] = model._vq_star._ema_w.detach().numpy()
        out["post_cluster_size"] = model._vq_star._ema_cluster_size.detach().numpy()
    np.savez_compressed(os.path.join(G.OUT, "sd_" + name + ".npz"), **{k: v.numpy() for k, v in sd_before.items()})
    if mode == "train":
        for k, p in model.named_parameters():
            out["grad_" + k] = p.grad.numpy() if p.grad is not None else np.zeros((0,), np.float32)
    np.savez_compressed(os.path.join(G.OUT, name + ".npz"), **out)
    print("wrote", name, "loss", float(out["loss"]), "perplexity", float(out["perplexity"]), "codes used", len(np.unique(out["tap_vq_indices"])))


def main():
    G.install_shims()
    mb = importlib.import_module("star.models.mae_base")
    VQ = importlib.import_module("star.models.VQSTAR")
    gen_quantizer(mb)
    gen_model_case(VQ, "vq_T1_comp_train", 1, "complement", B=1, A=5, num_agent=[5], S=32, mode="train", seed=21, K=64, E=32, cc=0.25)
    gen_model_case(VQ, "vq_T3_comp_infer", 3, "complement", B=2, A=5, num_agent=[5, 3], S=32, mode="infer", seed=22, K=96, E=16, cc=0.25)
    gen_model_case(VQ, "vq_T2_rand_train", 2, "random", B=1, A=5, num_agent=[4], S=32, mode="train", seed=23, K=32, E=64, cc=0.5, mask_ratio=0.5)
    gen_model_case(VQ, "vqema_T2_comp_train", 2, "complement", B=1, A=5, num_agent=[5], S=32, mode="train", seed=24, K=48, E=32, cc=0.25, decay=0.9)
    gen_model_case(VQ, "vqema_T1_comp_infer", 1, "complement", B=1, A=5, num_agent=[3], S=32, mode="infer", seed=25, K=48, E=32, cc=0.25, decay=0.9)


if __name__ == "__main__":
    main()
