"""SURVEY 8 row a15 -- the optimiser half of the training step (reference: completion/train_completion.py:124-125,
star/utils/misc.py:257-271, timm 0.3.2 add_weight_decay): star_amd.optim.FlatAdamW (one star_adamw launch over the flat
buffers, timm decay mask, upstream loss scale applied in the kernel) against torch.optim.AdamW over timm's two parameter
groups, betas (0.9, 0.95), stepping a model several times; and the state_dict() interchange with torch's layout (what the
reference checkpoint's `optimizer_state_dict` holds)."""
from functools import partial

import numpy as np
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from oracle import star_oracle as O
    from star_amd.models.multiagent_mae import AmortizedIndivMMAEViT
    from star_amd.optim import FlatAdamW, add_weight_decay
    from star_amd.synthetic import make_structured_batch
    from star_amd.utils.CoModule import CoModule

DEV = "cuda"
DIMS = dict(patch_size=8, in_chans=13, embed_dim=64, depth=2, num_heads=2, decoder_embed_dim=48, decoder_depth=1, decoder_num_heads=3,
            mlp_ratio=4, norm_layer=partial(nn.LayerNorm, eps=1e-6))
ODIMS = dict(patch=8, chans=13, dim=64, depth=2, heads=2, ddim=48, ddepth=1, dheads=3)


def _models(T, mask):
    torch.manual_seed(0)
    hip = AmortizedIndivMMAEViT(img_size=64, time_stamp=T, mask_method=mask, **DIMS)
    with torch.no_grad():
        hip.decoder_temp_embed.normal_(std=0.02)
    sd = {k: v.detach().clone() for k, v in hip.state_dict().items()}
    ref = AmortizedIndivMMAEViT(img_size=64, time_stamp=T, mask_method=mask, **DIMS)      # parameter container on the CPU: names, order, requires_grad
    ref.load_state_dict(sd)
    return hip.to(DEV).set_compute_dtype("fp32"), ref, sd


def _inputs(data):
    bev = data["bev_seq"].squeeze(1).permute(0, 3, 1, 2)
    nxt = data["bev_seq_next"]
    if nxt.dim() > 2:
        nxt = nxt.permute(0, 1, 4, 2, 3)
    return bev, nxt


@pytest.mark.parametrize("T,mask", [(2, "complement"), (3, "random")])
def test_flat_adamw_steps_match_torch_adamw_fed_the_same_gradients(T, mask):
    """Five steps with a changing lr and a NON-UNIT upstream loss scale (the gscale branch): after every step every parameter
    equals torch.optim.AdamW(add_weight_decay(model, 0.05), betas=(0.9, 0.95)) fed the gradients the HIP backward produced,
    times the scale.  Exercises the decay mask (wd_mask branch), the flat-buffer index mapping, bias correction and the
    rule that parameters without a gradient (cls_token, temp_embed, ...; reference Q5) are never touched."""
    hip, ref, sd = _models(T, mask)
    opt = FlatAdamW(hip, lr=1e-3, betas=(0.9, 0.95), weight_decay=0.05)
    topt = torch.optim.AdamW(add_weight_decay(ref, 0.05), lr=1e-3, betas=(0.9, 0.95))
    assert [g["weight_decay"] for g in topt.param_groups] == [0.0, 0.05]
    hp, rp = dict(hip.named_parameters()), dict(ref.named_parameters())
    data = make_structured_batch(2, 3, T, size=64, seed=7, device=DEV)
    bev, nxt = _inputs(data)
    scales = [1.0, 3.0, 0.5, 65536.0, 2.0]
    for it, scale in enumerate(scales):
        lr = 1e-3 * (it + 1) / 3.0
        for o in (opt, topt):
            for g in o.param_groups:
                g["lr"] = lr
        torch.manual_seed(40 + it)
        loss = hip(bev, nxt, None, data["trans_matrices"], data["num_agent"], 2, mask_ratio=0.6)[0]
        (loss * scale).backward()
        touched = set()
        for n, v in hip._grad_views.items():
            rp[n].grad = (v.detach().cpu() * scale).clone()           # what the scaled loss hands a per-tensor optimizer
            touched.add(n)
        for n in rp:
            if n not in touched:
                rp[n].grad = None
        opt.step()
        topt.step()
        for n in rp:
            torch.testing.assert_close(hp[n].detach().cpu(), rp[n].detach(), rtol=2e-6, atol=2e-8, msg=lambda s, n=n, it=it: "%s step %d: %s" % (n, it, s))
    for n in ("cls_token", "temp_embed"):          # never received a gradient: bit-identical to the initial value
        assert torch.equal(hp[n].detach().cpu(), sd[n]), n
    # decayed vs not decayed went the right way round: a 1-D tensor and a matrix both moved, by different rules
    assert not torch.equal(hp["blocks.0.norm1.weight"].detach().cpu(), sd["blocks.0.norm1.weight"])

    # ---- state_dict(): torch layout, both directions --------------------------------------------------------------
    s_hip, s_t = opt.state_dict(), topt.state_dict()
    assert [g["params"] for g in s_hip["param_groups"]] == [g["params"] for g in s_t["param_groups"]]
    assert [g["weight_decay"] for g in s_hip["param_groups"]] == [0.0, 0.05]
    assert sorted(s_hip["state"]) == sorted(s_t["state"])                      # the same parameter indices carry state
    for idx, st in s_t["state"].items():
        assert float(s_hip["state"][idx]["step"]) == float(st["step"]) == len(scales)
        torch.testing.assert_close(s_hip["state"][idx]["exp_avg"].cpu(), st["exp_avg"], rtol=2e-5, atol=1e-10)
        torch.testing.assert_close(s_hip["state"][idx]["exp_avg_sq"].cpu(), st["exp_avg_sq"], rtol=2e-5, atol=1e-14)
    # FlatAdamW -> torch: a fresh torch optimizer resumes from our state and takes the same next step
    ref2 = AmortizedIndivMMAEViT(img_size=64, time_stamp=T, mask_method=mask, **DIMS)
    ref2.load_state_dict({k: v.detach().cpu() for k, v in hip.state_dict().items()})
    topt2 = torch.optim.AdamW(add_weight_decay(ref2, 0.05), lr=1e-3, betas=(0.9, 0.95))
    topt2.load_state_dict({"state": {i: {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in st.items()} for i, st in s_hip["state"].items()},
                           "param_groups": s_hip["param_groups"]})
    # torch -> FlatAdamW: a fresh fused optimizer on a fresh model resumes from torch's state
    hip3, _, _ = _models(T, mask)
    hip3.load_state_dict({k: v.detach().cpu() for k, v in hip.state_dict().items()})
    opt3 = FlatAdamW(hip3, lr=1e-3, betas=(0.9, 0.95), weight_decay=0.05)
    opt3.load_state_dict(s_t)
    assert opt3.step_count == len(scales)
    torch.testing.assert_close(opt3.exp_avg, opt.exp_avg, rtol=2e-5, atol=1e-10)
    torch.testing.assert_close(opt3.exp_avg_sq, opt.exp_avg_sq, rtol=2e-5, atol=1e-14)
    rp2, hp3 = dict(ref2.named_parameters()), dict(hip3.named_parameters())
    torch.manual_seed(99)
    loss = hip(bev, nxt, None, data["trans_matrices"], data["num_agent"], 2, mask_ratio=0.6)[0]
    loss.backward()
    torch.manual_seed(99)
    loss3 = hip3(bev, nxt, None, data["trans_matrices"], data["num_agent"], 2, mask_ratio=0.6)[0]
    loss3.backward()
    assert loss.item() == loss3.item()
    for n, v in hip._grad_views.items():
        rp2[n].grad = v.detach().cpu().clone()
    lr = topt2.param_groups[0]["lr"]
    opt.param_groups[0]["lr"] = opt3.param_groups[0]["lr"] = lr
    opt.step(), topt2.step(), opt3.step()
    for n in rp2:
        torch.testing.assert_close(hp[n].detach().cpu(), rp2[n].detach(), rtol=2e-6, atol=2e-8, msg=lambda s, n=n: n + ": " + s)
        torch.testing.assert_close(hp3[n].detach().cpu(), hp[n].detach().cpu(), rtol=2e-6, atol=2e-8, msg=lambda s, n=n: n + ": " + s)


def test_training_trajectory_matches_oracle_with_torch_adamw():
    """End to end: four fused steps (lr schedule, forward, backward, star_adamw) through CoModule against the CPU oracle stepped
    by torch.optim.AdamW over timm's groups on ITS OWN gradients, same inputs and mask noise.  AdamW's normalised update turns a
    1e-6 relative gradient difference into a 1e-6 relative update difference, except where |g| ~ eps; so: losses within 1e-5
    at every step, the whole parameter update (p_4 - p_0) within 1 % in norm, and 99.9 % of its elements within 2e-5."""
    T, mask, B, A = 2, "complement", 2, 3
    hip, _, sd = _models(T, mask)
    opt = FlatAdamW(hip, lr=1e-3, betas=(0.9, 0.95), weight_decay=0.05)
    cm = CoModule(hip, opt, "ind_mae")
    osd = O.trainable(sd)
    topt = torch.optim.AdamW(O.weight_decay_groups(list(osd.items()), 0.05), lr=1e-3, betas=(0.9, 0.95))
    cfg = O.make_cfg(img=64, T=T, mask=mask, **ODIMS)
    hip.train()
    for it in range(4):
        data = make_structured_batch(B, A, T, size=64, seed=20 + it, device="cpu")
        noise = torch.rand(A * B, 64, generator=torch.Generator().manual_seed(300 + it))
        hip._noise_override = noise.to(DEV)
        lr = 1e-3 * (it + 1) / 4
        opt.param_groups[0]["lr"] = lr
        for g in topt.param_groups:
            g["lr"] = lr
        loss, _, _ = cm.step_mae_completion({k: (v.to(DEV) if torch.is_tensor(v) and k != "trans_matrices" else v) for k, v in data.items()},
                                            B, 0.75, trainable=True)
        bev, nxt = _inputs(data)
        ol = O.run(osd, cfg, bev.contiguous(), nxt.contiguous(), data["trans_matrices"].numpy(), data["num_agent"].numpy(), B, noise=noise)[0]
        topt.zero_grad()
        ol.backward()
        topt.step()
        assert abs(loss - ol.item()) < 1e-5, (it, loss, ol.item())
    hp = dict(hip.named_parameters())
    num = den = 0.0
    bad = tot = 0
    for k, v in osd.items():
        if not v.requires_grad:
            continue
        mine, ref, p0 = hp[k].detach().cpu(), v.detach(), sd[k]
        num += float((mine - ref).pow(2).sum())
        den += float((ref - p0).pow(2).sum())
        bad += int(((mine - ref).abs() > 2e-5).sum())
        tot += ref.numel()
    assert den > 0 and np.sqrt(num / den) < 1e-2, np.sqrt(num / den)
    assert bad <= 1e-3 * tot, (bad, tot)
