#!/usr/bin/env python3
"""Measure what the bf16 throughput mode does to decisions / IoU / gradients on the trained-weight fixtures and at full size
(the numbers the bounds in tests/test_trained_gpu.py are set from).  GPU box only."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from golden_util import MODEL_CASES, TRAINED_CASES, Case  # noqa: E402
from oracle import star_oracle as O  # noqa: E402
from test_model_gpu import build, run  # noqa: E402


def main():
    for name in TRAINED_CASES + MODEL_CASES:
        c = Case(name)
        gp = c.tap("pred")
        margin = (gp[:, 1] - gp[:, 0]).abs().numpy()
        line = "%-32s |margin| med %.3f  <0.25: %.4f  <1e-3: %.5f" % (name, np.median(margin), (margin < 0.25).mean(), (margin < 1e-3).mean())
        for dtype in ("fp32", "bf16"):
            m = build(c, dtype)
            loss, result, latent, ind = run(m, c)
            flips = ind.cpu().numpy() != c.ind
            meter = O.IoUMeter()
            meter.add_batch(result.cpu(), c.teacher)
            iou = meter.average()
            pe = (m._taps["pred"].cpu() - gp).abs().max().item()
            line += " | %s: flips %d/%d (%.2e) max margin of a flip %.4f, logit err %.2e, loss err %.2e, fused diff %d, iou d %.2e" % (
                dtype, flips.sum(), flips.size, flips.mean(), margin[flips].max() if flips.any() else 0.0, pe, abs(loss.item() - c.loss),
                int((result.cpu().numpy() != c.result).sum()), np.abs(iou - c.iou).max())
            if c.train:
                loss.backward()
                params = dict(m.named_parameters())
                errs = sorted(((((params[k].grad.cpu() - g).norm() / (g.norm() + 1e-12)).item(), k, g.norm().item()) for k, g in c.grads().items()
                               if g.numel() and g.norm() > 1e-7), reverse=True)
                line += ", worst grads " + "; ".join("%s %.3e (|g| %.2e)" % (k, e, n) for e, k, n in errs[:4])
                gs = {k: g for k, g in c.grads().items() if g.numel()}
                gall = torch.sqrt(sum(g.pow(2).sum() for g in gs.values())).item()
                eall = torch.sqrt(sum((params[k].grad.cpu() - g).pow(2).sum() for k, g in gs.items())).item()
                emax = max((params[k].grad.cpu() - g).norm().item() for k, g in gs.items())
                line += ", WHOLE-GRADIENT rel err %.3e, worst tensor abs err / |g_all| %.3e" % (eall / gall, emax / gall)
        print(line, flush=True)


def rounded_weights(name="trained_small_T1_comp_train"):
    """How much of the bf16-mode gradient error is the rounding of the WEIGHTS to bf16 (inherent to bf16 MFMA operands)?
    The oracle is evaluated at the bf16-rounded matrices (fp32 arithmetic); what is left against HIP bf16 is activation rounding."""
    c = Case(name)
    cfg = O.make_cfg(img=c.S, T=c.T, mask=c.mask, **c.dims, **c.flags)
    sd_r = {k: (v.bfloat16().float() if v.dim() >= 2 and k.endswith(".weight") else v.clone()) for k, v in c.sd.items()}
    osd = O.trainable(sd_r)
    loss, result, latent, ind, pred = O.run(osd, cfg, c.bev, c.next, c.trans, c.num_agent, c.B, mask_ratio=c.mask_ratio, infer=False, noise=c.noise)
    loss.backward()
    m = build(c, "bf16")
    hl, hr, hlat, hind = run(m, c)
    hl.backward()
    params = dict(m.named_parameters())
    ref = {k: g for k, g in c.grads().items() if g.numel()}
    gall = torch.sqrt(sum(g.pow(2).sum() for g in ref.values())).item()
    e_ref = torch.sqrt(sum((params[k].grad.cpu() - g).pow(2).sum() for k, g in ref.items())).item()
    e_rw = torch.sqrt(sum((params[k].grad.cpu() - osd[k].grad).pow(2).sum() for k in ref)).item()
    e_w = torch.sqrt(sum((osd[k].grad - g).pow(2).sum() for k, g in ref.items())).item()
    print("%s: whole-gradient error / |g_all|: HIP bf16 vs reference %.3e; oracle(bf16-rounded weights) vs reference %.3e; "
          "HIP bf16 vs oracle(bf16-rounded weights) %.3e; logits HIP vs oracle(rw) %.3e, oracle(rw) vs reference %.3e"
          % (name, e_ref / gall, e_w / gall, e_rw / gall, (m._taps["pred"].cpu() - pred.detach()).abs().max().item(),
             (pred.detach() - c.tap("pred")).abs().max().item()), flush=True)


def full_size(T=1, steps=1500, B=2, A=5, nb=16):
    """Train amortized_ind_patch8_light with the HIP bf16 path on the structured law, then compare one held-out 3-agent scene
    against the CPU oracle on the trained weights: HIP fp32 and HIP bf16."""
    import time

    from star_amd.models import multiagent_mae as mm
    from star_amd.optim import FlatAdamW
    from star_amd.synthetic import make_structured_batch
    from star_amd.utils.CoModule import CoModule

    torch.manual_seed(0)
    model = mm.amortized_ind_patch8_light(time_stamp=T, mask_method="complement").to("cuda")
    opt = FlatAdamW(model, lr=1e-3, betas=(0.9, 0.95), weight_decay=0.05)
    cm = CoModule(model, opt, "ind_mae")
    model.train()
    batches = [make_structured_batch(B, A, T, seed=100 + i) for i in range(nb)]
    t0 = time.time()
    for it in range(steps):
        opt.param_groups[0]["lr"] = 1.5e-3 * min(1.0, (it + 1) / 40.0) * (0.1 + 0.9 * 0.5 * (1 + np.cos(np.pi * it / steps)))
        torch.manual_seed(1000 + it)
        loss, _, _ = cm.step_mae_completion(batches[it % nb], B, 0.75, trainable=True)
        if it % 100 == 0 or it == steps - 1:
            print("  full-size step %3d loss %.4f" % (it, loss), flush=True)
    print("  trained in %.1f s" % (time.time() - t0), flush=True)
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    data = make_structured_batch(1, 3, T, seed=999, device="cpu")
    bev = data["bev_seq"].squeeze(1).permute(0, 3, 1, 2).contiguous()
    nxt = data["bev_seq_next"].permute(0, 1, 4, 2, 3).contiguous() if T > 1 else data["bev_seq_next"]
    teacher = data["bev_seq_teacher"].squeeze(1).permute(0, 3, 1, 2).contiguous()
    noise = torch.rand(3, 1024, generator=torch.Generator().manual_seed(5))
    cfg = O.make_cfg(T=T, mask="complement")
    osd = O.trainable(sd)
    t0 = time.time()
    ol, ores, olat, oind, opred = O.run(osd, cfg, bev, nxt, data["trans_matrices"].numpy(), data["num_agent"].numpy(), 1, noise=noise)
    ol.backward()
    print("  oracle fwd+bwd %.1f s, loss %.5f" % (time.time() - t0, ol.item()), flush=True)
    om = O.IoUMeter()
    om.add_batch(ores, teacher)
    margin = (opred[:, 1] - opred[:, 0]).abs().detach().numpy()
    print("  |margin| median %.3f, <0.25: %.5f, <0.05: %.5f, occupied decisions %.4f" % (np.median(margin), (margin < 0.25).mean(), (margin < 0.05).mean(), oind.float().mean()))
    for dtype in ("fp32", "bf16"):
        m2 = mm.amortized_ind_patch8_light(time_stamp=T, mask_method="complement")
        m2.load_state_dict(sd)
        m2 = m2.to("cuda").set_compute_dtype(dtype)
        m2.train()
        m2._noise_override = noise.to("cuda")
        m2._taps = {}
        loss, result, latent, ind = m2(bev.cuda(), nxt.cuda(), None, data["trans_matrices"], data["num_agent"].cuda(), 1)
        loss.backward()
        flips = ind.cpu().numpy() != oind.numpy()
        meter = O.IoUMeter()
        meter.add_batch(result.cpu(), teacher)
        pe = (m2._taps["pred"].cpu() - opred.detach()).abs()
        params = dict(m2.named_parameters())
        errs = sorted(((((params[k].grad.cpu() - v.grad).norm() / (v.grad.norm() + 1e-12)).item(), k, v.grad.norm().item()) for k, v in osd.items()
                       if v.grad is not None and v.grad.norm() > 1e-9), reverse=True)
        gall = torch.sqrt(sum(v.grad.pow(2).sum() for v in osd.values() if v.grad is not None)).item()
        eabs = max((params[k].grad.cpu() - v.grad).norm().item() for k, v in osd.items() if v.grad is not None)
        eall = torch.sqrt(sum((params[k].grad.cpu() - v.grad).pow(2).sum() for k, v in osd.items() if v.grad is not None)).item()
        print("     WHOLE-GRADIENT rel err %.3e" % (eall / gall))
        print("     global grad norm %.3e, worst per-tensor abs err / global norm %.3e" % (gall, eabs / gall))
        print("  %s: loss err %.2e, logit err max %.3e (rel to max |logit| %.3e, rms %.3e), flips %d/%d (%.2e), max margin of a flip %.4f, fused diff %d, iou %s vs %s"
              % (dtype, abs(loss.item() - ol.item()), pe.max().item(), pe.max().item() / opred.abs().max().item(), pe.pow(2).mean().sqrt().item(),
                 flips.sum(), flips.size, flips.mean(), margin[flips].max() if flips.any() else 0.0,
                 int((result.cpu().numpy() != ores.numpy()).sum()), meter.average(), om.average()))
        print("     worst grads: " + "; ".join("%s %.3e (|g| %.2e)" % (k, e, n) for e, k, n in errs[:6]))
        print("     median grad rel err %.3e" % np.median([e for e, _, _ in errs]), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "rw":
        for n in ("trained_small_T1_comp_train", "trained_small_T2_comp_train", "small_T2_comp_train"):
            rounded_weights(n)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "full":
        full_size(T=int(sys.argv[2]) if len(sys.argv) > 2 else 1)
        sys.exit(0)
    main()
