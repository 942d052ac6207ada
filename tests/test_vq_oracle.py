"""The VQ-STAR part of the CPU oracle against golden vectors captured from the reference's own quantiser and
VQSTARViT (tests/golden/gen_golden_vq.py).  No GPU."""
import numpy as np
import pytest
import torch

from golden_util import GOLD
from oracle import star_oracle as O
from vq_util import VQ_CASES, VQ_DIMS, VQ_EMA_CASES, VQCase


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_quantizer_alone(tag):
    z = np.load(GOLD + "/vq_quantizer.npz")
    x = torch.from_numpy(z[tag + "_x"]).requires_grad_(True)
    cb = torch.from_numpy(z[tag + "_codebook"]).requires_grad_(True)
    loss, q, perp, idx = O.vector_quantize(x, cb, float(z[tag + "_cc"]))
    assert np.array_equal(idx.numpy(), z[tag + "_idx"])             # incl. the duplicated code: first minimum
    assert abs(loss.item() - float(z[tag + "_loss"])) < 1e-6 and abs(perp.item() - float(z[tag + "_perplexity"])) < 1e-5
    assert np.array_equal(q.detach().numpy(), z[tag + "_quantized"])
    (loss + (q * torch.from_numpy(z[tag + "_w"])).sum()).backward()
    torch.testing.assert_close(x.grad, torch.from_numpy(z[tag + "_dx"]), rtol=1e-6, atol=1e-8)
    torch.testing.assert_close(cb.grad, torch.from_numpy(z[tag + "_dcodebook"]), rtol=1e-5, atol=1e-8)


@pytest.mark.parametrize("name", VQ_CASES + VQ_EMA_CASES)
def test_vq_model_case(name):
    c = VQCase(name)
    cfg = O.make_cfg(img=c.S, T=c.T, mask=c.mask, inter=c.E, **VQ_DIMS)
    cfg.update(vq=True, commitment=c.cc, vq_decay=c.decay)
    sd = O.trainable(c.sd) if c.train else c.sd
    taps = {}
    with torch.set_grad_enabled(c.train):
        loss, result, latent, ind, pred = O.run(sd, cfg, c.bev, c.next, c.trans, c.num_agent, c.B, mask_ratio=c.mask_ratio,
                                                infer=not c.train, noise=c.noise, taps=taps)
    assert torch.equal(taps["ids_restore"], c.tap("ids_restore")) and torch.equal(taps["mask"], c.tap("mask"))
    torch.testing.assert_close(latent.detach(), c.tap("latent"), rtol=2e-5, atol=2e-5)
    safe = c.z["vq_margin"] > 1e-4
    assert (taps["vq_indices"].numpy() == c.z["tap_vq_indices"])[safe].all() and safe.mean() > 0.99
    if safe.all():
        torch.testing.assert_close(taps["pred"].detach(), c.tap("pred"), rtol=2e-5, atol=2e-5)
        assert abs(loss.item() - c.loss) < 1e-6 and abs(taps["perplexity"].item() - c.perplexity) < 1e-5
        gp = c.tap("pred")
        ok = (gp[:, 1] - gp[:, 0]).abs().numpy() > 1e-4
        assert (ind.numpy() == c.ind)[ok].all()
        if ok.all():
            assert np.array_equal(result.numpy(), c.result)
        if c.train:
            loss.backward()
            for k, g in c.grads().items():
                if g.numel() == 0:
                    assert sd[k].grad is None, k
                else:
                    torch.testing.assert_close(sd[k].grad, g, rtol=1e-4, atol=1e-6, msg=lambda m: k + ": " + m)
        if c.decay > 0:     # EMA codebook: the state the reference holds after the step (unchanged in eval mode)
            ncb, nw, ncs = taps["vq_state"]
            torch.testing.assert_close(ncs, torch.from_numpy(c.z["post_cluster_size"]), rtol=1e-6, atol=1e-7)
            torch.testing.assert_close(nw, torch.from_numpy(c.z["post_ema_w"]), rtol=1e-6, atol=1e-7)
            torch.testing.assert_close(ncb, torch.from_numpy(c.z["post_embedding"]), rtol=1e-5, atol=1e-7)
            if not c.train:
                assert torch.equal(ncb, c.sd["_vq_star._embedding.weight"]) and torch.equal(ncs, c.sd["_vq_star._ema_cluster_size"])
