"""The CPU oracle against golden vectors captured from the reference (tests/golden/gen_golden.py).
Runs without a GPU.  Integer/index outputs bit-exact; floating point within 2e-5 (the oracle and the
reference run the same fp32 torch ops, differences come only from op grouping)."""
import numpy as np
import pytest
import torch

from golden_util import GOLD, MODEL_CASES, TRAINED_CASES, Case, unpack
from oracle import c_oracle as OC
from oracle import star_oracle as O

FP_TOL = dict(rtol=2e-5, atol=2e-5)


def test_pos_embed_tables():
    z = np.load(GOLD + "/standalone.npz")
    for key in ("pos_384_32", "pos_256_32", "pos_32_32", "pos_768_16", "pos_48_4"):
        _, d, g = key.split("_")
        mine = O.sincos_pos_embed(int(d), int(g)).astype(np.float32)
        assert np.array_equal(mine, z[key]), key


def test_lr_schedule():
    z = np.load(GOLD + "/standalone.npz")
    mine = np.array([O.lr_at(float(e), 1e-3, 1e-5, 4, 30) for e in z["lr_epochs"]])
    assert np.array_equal(mine, z["lr_values"])


def test_patchify_roundtrip_and_probe():
    z = np.load(GOLD + "/standalone.npz")
    x = torch.arange(2 * 13 * 256 * 256, dtype=torch.float32).reshape(2, 13, 256, 256) % 251
    p = O.patchify(x, 8)
    assert np.array_equal(p[:, ::97, ::53].numpy(), z["patchify_probe"])
    assert torch.equal(O.unpatchify(p, 8), x)


def test_occupancy_iou():
    z = np.load(GOLD + "/standalone.npz")
    pred = unpack(z["iou_pred_bits"], z["iou_shape"])
    tgt = unpack(z["iou_tgt_bits"], z["iou_shape"])
    m = O.IoUMeter()
    for b in range(pred.shape[0]):
        m.add_batch(torch.from_numpy(pred[b]), torch.from_numpy(tgt[b]))
    assert np.array_equal(np.asarray(m.per_batch), z["iou_every_batch"])
    assert np.array_equal(m.average(), z["iou_avg"])


def test_warp_and_fusion_full_resolution_bit_exact():
    z = np.load(GOLD + "/warp_full256.npz")
    ind = unpack(z["ind_bits"], z["shape"])
    gt = unpack(z["gt_bits"], z["shape"])
    trans, na = z["trans"], z["num_agent"]
    B = trans.shape[0]
    A = ind.shape[0] // B
    x = ind.reshape(A, B, *ind.shape[1:])
    single = unpack(z["single_bits"], z["single_shape"])
    for n, (b, i, j) in enumerate(z["pairs"]):
        assert np.array_equal(O.warp_map(x[j, b], trans[b, i, j]), single[n]), (b, i, j)
    assert np.array_equal(O.late_fusion(ind, trans, na, B), unpack(z["late_bits"], z["shape"]))
    assert np.array_equal(O.late_fusion(ind, trans, na, B, ego_maps=gt), unpack(z["ego_bits"], z["shape"]))
    assert np.array_equal(O.late_fusion(ind, trans.astype(np.float32), na, B), unpack(z["late32_bits"], z["shape"]))


@pytest.mark.parametrize("name", MODEL_CASES + TRAINED_CASES)
def test_model_case(name):
    c = Case(name)
    cfg = O.make_cfg(img=c.S, T=c.T, mask=c.mask, **c.dims, **c.flags)
    sd = O.trainable(c.sd) if c.train else c.sd
    taps = {}
    with torch.set_grad_enabled(c.train):
        loss, result, latent, ind, pred = O.run(sd, cfg, c.bev, c.next, c.trans, c.num_agent, c.B,
                                                mask_ratio=c.mask_ratio, infer=not c.train, noise=c.noise, taps=taps)
    # index work: bit-exact
    assert torch.equal(taps["ids_restore"], c.tap("ids_restore"))
    assert torch.equal(taps["mask"], c.tap("mask"))
    # floating point stages
    for k in ("tok_patch", "enc_blk0", "enc_norm", "dec_embed", "dec_in", "dec_norm", "pred"):
        torch.testing.assert_close(taps[k].detach(), c.tap(k).reshape(taps[k].shape), **FP_TOL, msg=lambda m: k + ": " + m)
    torch.testing.assert_close(latent.detach(), torch.from_numpy(c.z["latent"]), **FP_TOL)
    assert abs(loss.item() - c.loss) < 1e-6
    # occupancy decisions and fused maps: bit-exact wherever the golden margin is not a rounding tie
    gp = c.tap("pred")
    margin = (gp[:, 1] - gp[:, 0]).abs().numpy()
    safe = margin > 1e-4
    assert (ind.numpy() == c.ind)[safe].all()
    if safe.all():
        assert np.array_equal(result.numpy(), c.result)
        m = O.IoUMeter()
        m.add_batch(result, c.teacher)
        assert np.array_equal(m.average(), c.iou)
    if c.train:
        loss.backward()
        for k, g in c.grads().items():
            if g.numel() == 0:
                assert sd[k].grad is None, k
            else:
                torch.testing.assert_close(sd[k].grad, g, rtol=1e-4, atol=1e-6, msg=lambda m: k + ": " + m)


def test_rng_consumption_matches_reference():
    """Q3: one torch.rand(BA, L) (complement) or torch.rand(BA*T, L) (random) per forward, so seeding
    torch.manual_seed(123) reproduces the captured noise."""
    for name in ("tiny_T1_comp_train", "tiny_T3_rand_train"):
        c = Case(name)
        cfg = O.make_cfg(img=c.S, T=c.T, mask=c.mask, **c.dims, **c.flags)
        torch.manual_seed(123)
        taps = {}
        with torch.no_grad():
            O.run(c.sd, cfg, c.bev, c.next, c.trans, c.num_agent, c.B, mask_ratio=c.mask_ratio, taps=taps)
        assert torch.equal(taps["ids_restore"], c.tap("ids_restore"))


def test_c_oracle_warp_fusion_matches_reference_goldens():
    """The plain-C restatement (oracle/star_oracle_c.c) against the same full-resolution reference outputs."""
    z = np.load(GOLD + "/warp_full256.npz")
    ind = unpack(z["ind_bits"], z["shape"])
    gt = unpack(z["gt_bits"], z["shape"])
    trans, na = z["trans"], z["num_agent"]
    B = trans.shape[0]
    assert np.array_equal(OC.late_fusion(ind, trans, na, B), unpack(z["late_bits"], z["shape"]))
    assert np.array_equal(OC.late_fusion(ind, trans, na, B, ego_maps=gt), unpack(z["ego_bits"], z["shape"]))
    assert np.array_equal(OC.late_fusion(ind, trans.astype(np.float32), na, B), unpack(z["late32_bits"], z["shape"]))


def test_c_oracle_argsort_and_confusion():
    rng = np.random.RandomState(0)
    noise = rng.rand(7, 1024).astype(np.float32)
    noise[:, 100:110] = noise[:, 90:100]          # ties: stable order decides
    s, inv = OC.shuffle_ids(noise)
    ts, tinv = O.shuffle_ids(torch.from_numpy(noise))
    assert np.array_equal(s, ts.numpy()) and np.array_equal(inv, tinv.numpy())
    checked = 0
    for name in MODEL_CASES[:4]:                   # ids captured from the reference run
        z = np.load(GOLD + "/%s.npz" % name)
        if "noise" in z.files and "tap_ids_restore" in z.files:
            _, inv = OC.shuffle_ids(z["noise"].reshape(-1, z["noise"].shape[-1]))
            assert np.array_equal(inv.reshape(z["tap_ids_restore"].shape), z["tap_ids_restore"]), name
            checked += 1
    assert checked > 0
    pred = (rng.rand(3, 13, 64, 64) < 0.3).astype(np.float32)
    tgt = (rng.rand(3, 13, 64, 64) < 0.3).astype(np.float32)
    conf = OC.confusion(pred, tgt)
    want = np.array([[((tgt == t) & (pred == p)).sum() for p in (0, 1)] for t in (0, 1)])
    assert np.array_equal(conf, want)
