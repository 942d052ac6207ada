import sys, os
sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo")
os.chdir("/root/repo")
import numpy as np, torch
from joint_util import JOINT_CASES, JointCase
import test_joint_gpu as T
for name in JOINT_CASES:
    c = JointCase(name)
    m = T.build(c, "bf16")
    loss, result, mask, pred = T.run(m, c)
    t = m._taps
    feat_err = (t["feat"].cpu().reshape(c.tap("feat").shape) - c.tap("feat")).abs()
    dpos = m.decoder_pos_embed[0, 1:].cpu()
    fused = t["fused"].cpu() - dpos
    flips = ((fused - c.tap("fused")).abs() > 0.5).float().mean().item()
    pe = (pred.cpu() - c.pred).abs()
    margin = (c.pred[:, 1] - c.pred[:, 0]).abs().numpy()
    dec = (result.cpu().numpy() != c.result)
    line = "%s feat max %.3e rms %.3e (ref rms %.3e) | fused flips %.4f | logits max %.3e rms %.3e (ref rms %.3e, max %.2f) | loss %.5f vs %.5f | decisions differ %.4f, max margin %.3f" % (
        name, feat_err.max(), feat_err.pow(2).mean().sqrt(), c.tap("feat").pow(2).mean().sqrt(), flips, pe.max(), pe.pow(2).mean().sqrt(), c.pred.pow(2).mean().sqrt(), c.pred.abs().max(),
        loss.item(), c.loss, dec.mean(), margin[dec].max() if dec.any() else 0)
    if c.train:
        loss.backward()
        params = dict(m.named_parameters())
        num = den = 0.0
        worst = (0, "")
        for k, g in c.grads().items():
            if g.numel() and params[k].grad is not None:
                e = float((params[k].grad.cpu() - g).norm()); n = float(g.norm())
                num += e * e; den += n * n
                if n > 0 and e / n > worst[0]: worst = (e / n, k)
        line += " | grad whole %.3f worst %.3f %s" % ((num / den) ** 0.5, worst[0], worst[1])
    print(line, flush=True)
