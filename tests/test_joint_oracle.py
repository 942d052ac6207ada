"""The oracle's restatement of the JOINT model (oracle/star_oracle.py::run_joint) against golden vectors
captured from the reference's AmortizedFusionMMAEViT + ConvPred / ConvPred16 (tests/golden/gen_golden_joint.py).  CPU only."""
import numpy as np
import pytest
import torch

from joint_util import JOINT_CASES, JointCase
from oracle import star_oracle as O


@pytest.mark.parametrize("name", JOINT_CASES)
def test_joint_case(name):
    c = JointCase(name)
    cfg = O.make_cfg(img=c.S, T=c.T, mask=c.mask, **c.dims)
    sd = O.trainable(c.sd) if c.train else c.sd
    taps = {}
    with torch.set_grad_enabled(c.train):
        loss, result, mask, pred = O.run_joint(sd, cfg, c.bev, c.next, c.teacher, c.trans, c.num_agent, c.B, mask_ratio=c.mask_ratio,
                                               noise=c.noise, training=c.train, taps=taps)
    assert torch.equal(mask, torch.from_numpy(c.z["mask"]))
    torch.testing.assert_close(taps["feat"].detach().reshape(c.tap("feat").shape), c.tap("feat"), rtol=2e-5, atol=2e-5)
    # the thresholded latent is {0,1} (passthrough slots excepted): identical wherever the summed features are not within 1e-4 of 0.5
    assert (taps["fused"].detach() - c.tap("fused")).abs().max() < 1e-4
    torch.testing.assert_close(pred.detach(), c.pred, rtol=1e-4, atol=1e-4)
    assert abs(loss.item() - c.loss) < 1e-5
    margin = (c.pred[:, 1] - c.pred[:, 0]).abs().numpy()
    assert (result.numpy() == c.result)[margin > 1e-4].all()
    if c.train:
        for k, v in taps["bn_stats"].items():       # running statistics after the training forward
            torch.testing.assert_close(v, c.after[k], rtol=1e-4, atol=1e-5, msg=lambda m, k=k: k + ": " + m)
        loss.backward()
        for k, g in c.grads().items():
            mine = sd[k].grad
            if g.numel() == 0:
                assert mine is None, k
            elif mine is None:          # reached only through thresholded (constant) slots: the reference holds exact zeros there
                assert float(g.abs().max()) == 0.0, k
            else:
                torch.testing.assert_close(mine, g, rtol=2e-3, atol=2e-6, msg=lambda m, k=k: k + ": " + m)
