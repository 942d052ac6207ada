"""Parity of BOTH compute modes on TRAINED weights (confident logits), where the occupancy decisions / fused maps / IoU that
the north star asks to be identical are meaningful -- the random-init fixtures of test_model_gpu.py sit at |margin| ~ 0.25.

Reference-pinned part: tests/golden/trained_*.npz, written by tests/golden/gen_golden_trained.py, which trains the IMPORTED
reference models on the CPU and captures their outputs.  Full-size part: amortized_ind_patch8_light trained here with the HIP
path, then HIP against the (pinned) CPU oracle on the trained weights.

What is asserted, with the values measured on MI355X (tools/measure_bf16_parity.py, profiles/r02_bf16_parity.log):

  fp32 mode  identical decisions, fused maps and IoU on every trained fixture; at full size identical except exact near-ties
             (round 3 run: 1 flip of 2.56 M voxels, at a reference margin of 4e-6; the bound asserted is margin < 1e-3);
             logits within 3e-4 (|logit| up to 15), gradients within 1e-4 relative.
  bf16x3     (inference; tests/test_bf16x3_gpu.py has the fixture assertions) the float32 mode's decisions: the same single
             near-tie flips at full size, logits within 6e-4.
  bf16 mode  NOT identical: bf16 storage of MFMA operands perturbs the logits by ~0.25 % rms (max 2.8 % of the largest
             logit), which flips voxels whose two logits nearly tie.  Measured: 0 ... 1.6e-4 of the voxels on the trained
             fixtures (every flipped voxel has a reference margin < 0.021), 3.8e-5 at full size (margin < 0.071, median margin
             12); IoU moves by <= 8.5e-4 (fixtures) / 1e-4 (full size).  Bounds below: flip fraction <= 3e-4, every flip at a
             reference margin < 0.1 (fixtures) / 0.2 (full size), IoU within 2e-3, fused maps differ on <= 5e-4 of the cells.
             Gradients: see _grad_checks -- the rounding of the weights to bf16 dominates (it moves a converged model's
             gradient by up to 21 % of its norm in exact arithmetic); against the oracle evaluated at the rounded weights the
             HIP path is within 3.5 % (whole vector).  At full size on trained weights: 2.3 % whole vector against the reference
             weights' gradient, per-tensor 1.1 ... 5.5 % (median 1.9 %); on the lightly trained model of test_model_gpu.py <= 1.3 %.
"""
import numpy as np
import pytest
import torch

from golden_util import TRAINED_CASES, Case

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from oracle import star_oracle as O
    from test_model_gpu import build, run

DEV = "cuda"


def _iou(result, teacher):
    m = O.IoUMeter()
    m.add_batch(result, teacher)
    return m.average()


@pytest.mark.parametrize("name", TRAINED_CASES)
def test_trained_fixture_fp32_mode_is_identical(name):
    c = Case(name)
    m = build(c, "fp32")
    loss, result, latent, ind = run(m, c)
    gp = c.tap("pred")
    torch.testing.assert_close(m._taps["pred"].cpu(), gp, rtol=2e-4, atol=2e-4)
    assert abs(loss.item() - c.loss) < 1e-5
    assert np.array_equal(ind.cpu().numpy(), c.ind)                       # every decision, no margin carve-out
    assert np.array_equal(result.cpu().numpy(), c.result)                  # fused maps
    assert np.array_equal(_iou(result.cpu(), c.teacher), c.iou)            # identical thresholded IoU
    if c.train:
        loss.backward()
        params = dict(m.named_parameters())
        for k, g in c.grads().items():
            if g.numel():
                torch.testing.assert_close(params[k].grad.cpu(), g, rtol=2e-3, atol=2e-6, msg=lambda s, k=k: k + ": " + s)


@pytest.mark.parametrize("name", TRAINED_CASES)
def test_trained_fixture_bf16_mode_bounded(name):
    c = Case(name)
    m = build(c, "bf16")
    loss, result, latent, ind = run(m, c)
    gp = c.tap("pred")
    err = (m._taps["pred"].cpu() - gp).abs()
    assert err.max().item() < 0.02 * gp.abs().max().item() + 0.01, err.max().item()     # measured <= 1.3 % of max |logit|
    assert abs(loss.item() - c.loss) < 1e-4                                            # measured <= 1.8e-5
    margin = (gp[:, 1] - gp[:, 0]).abs().numpy()
    flips = ind.cpu().numpy() != c.ind
    assert flips.mean() <= 3e-4, flips.mean()                                           # measured <= 1.6e-4
    assert not flips.any() or margin[flips].max() < 0.1, margin[flips].max()           # measured < 0.021: near-ties only
    assert (result.cpu().numpy() != c.result).mean() <= 5e-4                           # measured <= 2.1e-4
    assert np.abs(_iou(result.cpu(), c.teacher) - c.iou).max() < 2e-3                  # measured <= 8.5e-4
    # the fusion itself is exact: fusing the HIP decisions with the oracle reproduces the HIP output
    want = O.late_fusion(ind.cpu().numpy().astype(np.float32), c.trans, c.num_agent, c.B, ego_maps=None if c.train else c.bev.numpy())
    assert np.array_equal(result.cpu().numpy(), want)
    if c.train:
        loss.backward()
        mine = {k: p.grad.cpu() for k, p in m.named_parameters() if p.grad is not None}
        grads = {k: g for k, g in c.grads().items() if g.numel()}
        # (1) against the reference's gradient, whole vector (measured 19 % / 4.3 % / 0.34 %: see _grad_checks)
        # (2) against the oracle evaluated AT the bf16-rounded weights: activation rounding only (measured 3.5 % / 0.7 %)
        cfg = O.make_cfg(img=c.S, T=c.T, mask=c.mask, **c.dims, **c.flags)
        osd = O.trainable(_round_weights(c.sd))
        O.run(osd, cfg, c.bev, c.next, c.trans, c.num_agent, c.B, mask_ratio=c.mask_ratio, noise=c.noise)[0].backward()
        _grad_checks(mine, grads, {k: osd[k].grad for k in grads}, ref_bound=0.35, rw_bound=0.08)


def _round_weights(sd):
    """The weight matrices as the bf16 MFMA sees them (round to nearest even), everything else untouched."""
    return {k: (v.bfloat16().float() if v.dim() >= 2 and k.endswith(".weight") else v.clone()) for k, v in sd.items()}


def _whole(a, b, keys):
    return float(torch.sqrt(sum((a[k] - b[k]).pow(2).sum() for k in keys)))


def _grad_checks(mine, ref, ref_rw, ref_bound, rw_bound):
    """bf16 mode.  A converged model's gradient is a cancelling sum, so rounding the WEIGHTS to bf16 (inherent to bf16 MFMA
    operands) moves it by H.dW, which does not shrink with |g|: measured on trained_small_T1 the fp32 oracle evaluated at the
    bf16-rounded weights is 21 % (whole vector) off the reference gradient, and the HIP bf16 path 19 %.  Hence two checks:
    against the reference with the loose whole-vector bound, and against the oracle AT the rounded weights -- same function,
    only activation rounding left -- with the tight ones."""
    keys = list(ref)
    gall = float(torch.sqrt(sum(g.pow(2).sum() for g in ref.values())))
    assert _whole(mine, ref, keys) <= ref_bound * gall, (_whole(mine, ref, keys) / gall)
    gall_rw = float(torch.sqrt(sum(g.pow(2).sum() for g in ref_rw.values())))
    assert _whole(mine, ref_rw, keys) <= rw_bound * gall_rw, (_whole(mine, ref_rw, keys) / gall_rw)
    for k in keys:
        e = float((mine[k] - ref_rw[k]).norm())
        assert e <= 0.06 * float(ref_rw[k].norm()) + 0.03 * gall_rw, (k, e, float(ref_rw[k].norm()), gall_rw)


def _train_light(T, steps, B, A, nb, dtype="fp32", every=0):
    """-> (state dict, last loss, [loss at steps every-1, 2 every-1, ...])"""
    from star_amd.models import multiagent_mae as mm
    from star_amd.optim import FlatAdamW
    from star_amd.synthetic import make_structured_batch
    from star_amd.utils.CoModule import CoModule

    torch.manual_seed(0)
    model = mm.amortized_ind_patch8_light(time_stamp=T, mask_method="complement").to(DEV).set_compute_dtype(dtype)
    opt = FlatAdamW(model, lr=1e-3, betas=(0.9, 0.95), weight_decay=0.05)
    cm = CoModule(model, opt, "ind_mae")
    model.train()
    batches = [make_structured_batch(B, A, T, seed=100 + i) for i in range(nb)]
    trail = []
    for it in range(steps):
        opt.param_groups[0]["lr"] = 1.5e-3 * min(1.0, (it + 1) / 40.0) * (0.1 + 0.9 * 0.5 * (1 + np.cos(np.pi * it / steps)))
        torch.manual_seed(1000 + it)
        loss, _, _ = cm.step_mae_completion(batches[it % nb], B, 0.75, trainable=True)
        if every and (it + 1) % every == 0:
            trail.append(float(loss))
    return {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}, loss, trail


# Loss of the float32-mode training run below at steps 75, 150, ..., 1500 (seeds, data and schedule are fixed; measured on MI355X,
# round 3).  The run is what produces the weights the parity assertions are made on: pinning its trajectory keeps "the weights the
# test trains" the same object from build to build (float32 kernels differ between builds by summation order only: ~1e-6 per step).
TRAIN_TRAIL = [0.07595, 0.0575, 0.05314, 0.05452, 0.04438, 0.04278, 0.03844, 0.03464, 0.03448, 0.02964, 0.02652, 0.02283, 0.02334, 0.02181,
               0.02042, 0.0201, 0.01656, 0.01804, 0.01673, 0.01835]


def test_full_size_trained_model_both_modes_against_oracle():
    """amortized_ind_patch8_light (256 x 256 x 13, L = 1024) trained for 1500 float32-mode steps with the HIP path on the structured
    law (loss 0.81 -> 0.015, median logit margin 12, 1.4 % occupied decisions; the loss trajectory is pinned: TRAIN_TRAIL), then one
    held-out 3-agent scene: the CPU oracle on the trained weights against HIP fp32 and HIP bf16x3 (identical decisions / fused maps /
    IoU wherever the oracle's margin exceeds the logit error bound) and HIP bf16 (bounded, see module doc)."""
    from star_amd.models import multiagent_mae as mm
    from star_amd.synthetic import make_structured_batch

    T = 1
    sd, last, trail = _train_light(T, 1500, 2, 5, 16, dtype="fp32", every=75)
    print("TRAIN_TRAIL =", [round(v, 5) for v in trail])
    assert last < 0.03, last                      # it did learn (all-free prediction sits at 0.06)
    if TRAIN_TRAIL is not None:
        np.testing.assert_allclose(trail, TRAIN_TRAIL, rtol=0.03, atol=2e-4)
    data = make_structured_batch(1, 3, T, seed=999, device="cpu")
    bev = data["bev_seq"].squeeze(1).permute(0, 3, 1, 2).contiguous()
    nxt = data["bev_seq_next"]
    teacher = data["bev_seq_teacher"].squeeze(1).permute(0, 3, 1, 2).contiguous()
    noise = torch.rand(3, 1024, generator=torch.Generator().manual_seed(5))
    osd = O.trainable(sd)
    ol, ores, olat, oind, opred = O.run(osd, O.make_cfg(T=T, mask="complement"), bev, nxt, data["trans_matrices"].numpy(),
                                        data["num_agent"].numpy(), 1, noise=noise)
    ol.backward()
    opred = opred.detach()
    margin = (opred[:, 1] - opred[:, 0]).abs().numpy()
    assert np.median(margin) > 3.0 and 0.003 < oind.float().mean() < 0.1      # confident, and not "free everywhere"
    oiou = _iou(ores, teacher)
    ograds = {k: v.grad for k, v in osd.items() if v.grad is not None and v.grad.norm() > 1e-9}
    gall = float(torch.sqrt(sum(g.pow(2).sum() for g in ograds.values())))
    for dtype in ("fp32", "bf16x3", "bf16"):
        m = mm.amortized_ind_patch8_light(time_stamp=T, mask_method="complement")
        m.load_state_dict(sd)
        m = m.to(DEV).set_compute_dtype(dtype)
        m.train()
        m._noise_override = noise.to(DEV)
        m._taps = {}
        if dtype == "bf16x3":      # the inference mode: forward() without the autograd edge, the decisions' assertions of the float32 mode
            with torch.no_grad():
                loss, result, latent, ind = m(bev.to(DEV), nxt.to(DEV), None, data["trans_matrices"], data["num_agent"].to(DEV), 1)
        else:
            loss, result, latent, ind = m(bev.to(DEV), nxt.to(DEV), None, data["trans_matrices"], data["num_agent"].to(DEV), 1)
            loss.backward()
        pe = (m._taps["pred"].cpu() - opred).abs()
        flips = ind.cpu().numpy() != oind.numpy()
        params = dict(m.named_parameters())
        rel = {k: float((params[k].grad.cpu() - g).norm() / g.norm()) for k, g in ograds.items()} if dtype != "bf16x3" else {"-": 0.0}
        print("full size, %s: max |dlogit| %.2e, flips %d, max margin of a flip %.2e" % (
            dtype, pe.max().item(), int(flips.sum()), float(margin[flips].max()) if flips.any() else 0.0))
        if dtype in ("fp32", "bf16x3"):
            assert abs(loss.item() - ol.item()) < 1e-6
            assert pe.max().item() < 1e-3                                      # measured 1.2e-4 .. 2.6e-4 at |logit| <= 15
            # identical decisions wherever the oracle's margin exceeds the logit error bound; the weights come out of a bf16
            # training run on the GPU, so whether one of the 2.6 M voxels sits within 1e-3 of a tie differs from build to build
            assert not flips.any() or margin[flips].max() < 1e-3, (int(flips.sum()), float(margin[flips].max()))
            if not flips.any():
                assert np.array_equal(result.cpu().numpy(), ores.numpy())
                assert np.array_equal(_iou(result.cpu(), teacher), oiou)
            else:       # a handful of exact near-ties: the fusion of MY decisions must still be the oracle's fusion of them
                assert flips.sum() <= 20, int(flips.sum())
                want = O.late_fusion(ind.cpu().numpy().astype(np.float32), data["trans_matrices"].numpy(), data["num_agent"].numpy(), 1)
                assert np.array_equal(result.cpu().numpy(), np.asarray(want))
                assert np.abs(_iou(result.cpu(), teacher) - oiou).max() < 5e-4      # 3 fused voxels of ~36 k occupied ones
            assert max(rel.values()) < 1e-3, max(rel.items(), key=lambda kv: kv[1])     # measured 1.1e-5
        else:
            assert abs(loss.item() - ol.item()) < 1e-4                         # measured 1e-6
            assert pe.max().item() < 0.06 * opred.abs().max().item()           # measured 2.8 % of the largest |logit|
            assert pe.pow(2).mean().sqrt().item() < 0.01 * opred.pow(2).mean().sqrt().item()   # rms: measured 0.25 %
            assert flips.mean() <= 2e-4, flips.mean()                          # measured 3.8e-5
            assert not flips.any() or margin[flips].max() < 0.2                # measured 0.07 (median margin 12)
            assert (result.cpu().numpy() != ores.numpy()).mean() <= 5e-4       # measured 8e-5
            assert np.abs(_iou(result.cpu(), teacher) - oiou).max() < 1e-3     # measured 1e-4
            worst = max(rel.items(), key=lambda kv: kv[1])
            assert worst[1] < 0.10, worst                                      # measured 5.5 % (blocks.0.norm1.weight)
            assert float(np.median(list(rel.values()))) < 0.04                 # measured 1.9 %
            rsd = O.trainable(_round_weights(sd))
            O.run(rsd, O.make_cfg(T=T, mask="complement"), bev, nxt, data["trans_matrices"].numpy(), data["num_agent"].numpy(), 1,
                  noise=noise)[0].backward()
            mine = {k: params[k].grad.cpu() for k in ograds}
            _grad_checks(mine, ograds, {k: rsd[k].grad for k in ograds}, ref_bound=0.06, rw_bound=0.05)     # measured 2.3 % whole vector
