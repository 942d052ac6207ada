"""Compute dtype 'bf16x3' (STAR_BF16X3): bf16 hi/lo pairs, three bf16 MFMA products per multiply -- the inference mode that has
to reproduce the float32 path's DECISIONS (north star: identical thresholded occupancy IoU) at bf16-pipe speed.

Kernel level: star_layernorm_fwd / star_gemm / star_attn_fwd in the pair storage against float64 at float32-grade tolerances
(a pair carries 16 significant bits; a product drops the lo*lo term: 2^-16 relative per operand, measured ~3e-6 rms on outputs).
Model level: the trained fixtures captured from the imported reference (tests/golden/trained_*.npz) -- the same assertions as
test_trained_fixture_fp32_mode_is_identical: every decision, every fused cell, the IoU triple."""
import math

import numpy as np
import pytest
import torch

from golden_util import TRAINED_CASES, Case

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from oracle import star_oracle as O
    from star_amd import ops
    from star_amd._lib import ACT_GELU, ACT_NONE, ACT_RELU, BF16, BF16X3, GEMM_A_PLAIN
    from test_model_gpu import build, run

DEV = "cuda"


def _pairs(x):
    """float32 [R, C] -> bf16 [R, 2C] hi/lo pairs, through the library's own cast"""
    return ops.cast_split(x.contiguous())


def _unpair(p):
    c = p.shape[1] // 2
    return p[:, :c].double() + p[:, c:].double()


def test_cast_split_pairs_carry_16_bits():
    g = torch.Generator().manual_seed(0)
    x = (torch.randn(300, 200, generator=g) * torch.logspace(-3, 3, 200)).to(DEV)
    p = _pairs(x)
    assert p.dtype == torch.bfloat16 and tuple(p.shape) == (300, 400)
    assert torch.equal(p[:, :200], x.bfloat16())                                   # hi = round-to-nearest bf16
    rel = ((_unpair(p) - x.double()).abs() / x.double().abs().clamp_min(1e-30)).max().item()
    assert rel < 2.0 ** -16, rel


@pytest.mark.parametrize("M,N,K", [(300, 128, 64), (260, 144, 48), (130, 48, 72), (4096, 1152, 384), (1000, 384, 1536), (513, 32, 384), (700, 1664, 256)])
def test_gemm_pairs_against_float64(M, N, K):
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g).to(DEV)
    w = (torch.randn(N, K, generator=g) * K ** -0.5).to(DEV)
    bias = torch.randn(N, generator=g).to(DEV)
    res = torch.randn(M, N, generator=g).to(DEV)
    ap, wp = _pairs(a), _pairs(w)
    ref = a.double() @ w.double().t() + bias.double()
    scale = float(ref.abs().max())
    # plain Linear, pair output
    lp, _, _ = ops.gemm(ap, wp, bias=bias, dt=BF16X3)
    assert tuple(lp.shape) == (M, 2 * N)
    assert ((_unpair(lp) - ref).abs().max().item()) < 3e-5 * scale
    # residual epilogue, float32 output (proj / fc2)
    _, f32, _ = ops.gemm(ap, wp, bias=bias, res=res, want_lp=False, want_f32=True, dt=BF16X3)
    assert (f32.double() - (ref + res.double())).abs().max().item() < 3e-5 * scale
    # GELU epilogue (fc1)
    lp, _, _ = ops.gemm(ap, wp, bias=bias, act=ACT_GELU, dt=BF16X3)
    want = torch.nn.functional.gelu(ref)
    assert (_unpair(lp) - want).abs().max().item() < 3e-5 * scale
    # run-time epilogue: ReLU, gathered A rows, both outputs (compressor); scattered rows + two residual tables (decoder_embed)
    rows = torch.randint(-1, M, (M,), generator=g).int().to(DEV)
    lp, f32, _ = ops.gemm(ap, wp, bias=bias, act=ACT_RELU, a_rows=rows, want_f32=True, dt=BF16X3)
    az = torch.cat([a.double(), torch.zeros(1, K, device=DEV, dtype=torch.double)])[rows.long()]
    want = torch.relu(az @ w.double().t() + bias.double())
    assert (f32.double() - want).abs().max().item() < 3e-5 * scale and (_unpair(lp) - want).abs().max().item() < 3e-5 * scale
    perm = torch.randperm(M, generator=g).int().to(DEV)
    tab = torch.randn(7, N, generator=g).to(DEV)
    _, f32, _ = ops.gemm(ap, wp, bias=bias, c_rows=perm, out_rows=M, res=res, res_mod=M, res2=tab, res2_div=3, res2_mod=7, want_lp=False,
                         want_f32=True, dt=BF16X3)
    want = torch.empty_like(ref)
    m_idx = torch.arange(M, device=DEV)
    want[perm.long()] = ref + res.double()[perm.long()] + tab.double()[(m_idx // 3) % 7]
    assert (f32.double() - want).abs().max().item() < 3e-5 * scale
    # exact (0/1) plain-bf16 A operand: two products
    occ = (torch.rand(M, K, generator=g) < 0.1).to(DEV).bfloat16()
    _, f32, _ = ops.gemm(occ, wp, bias=bias, want_lp=False, want_f32=True, flags=GEMM_A_PLAIN, dt=BF16X3)
    want = occ.double() @ w.double().t() + bias.double()
    assert (f32.double() - want).abs().max().item() < 2e-5 * float(want.abs().max())


@pytest.mark.parametrize("D", [256, 384, 768])
def test_layernorm_pairs(D):
    g = torch.Generator().manual_seed(D)
    x = (torch.randn(1000, D, generator=g) * 3 + 0.5).to(DEV)
    gamma, beta = torch.randn(D, generator=g).to(DEV), torch.randn(D, generator=g).to(DEV)
    y, mean, rstd = ops.layernorm_fwd(x, gamma, beta, 1e-6, BF16X3)
    want = torch.nn.functional.layer_norm(x.double(), (D,), gamma.double(), beta.double(), 1e-6)
    assert tuple(y.shape) == (1000, 2 * D)
    assert (_unpair(y) - want).abs().max().item() < 2e-5 * float(want.abs().max())
    y16, _, _ = ops.layernorm_fwd(x, gamma, beta, 1e-6, BF16)
    assert torch.equal(y[:, :D], y16)                    # the hi halves are the bf16 kernel's output


@pytest.mark.parametrize("NB,L,H,hd", [(3, 1024, 12, 32), (2, 1024, 16, 16), (2, 342, 4, 32), (1, 100, 3, 64), (2, 77, 2, 16)])
def test_attention_pairs_against_float64(NB, L, H, hd):
    g = torch.Generator().manual_seed(L + hd)
    qkv = torch.randn(NB * L, 3 * H * hd, generator=g).to(DEV)
    qkv[:, : H * hd] *= 1.5                                            # sharper softmax rows
    out, lse = ops.attn_fwd(_pairs(qkv), NB, L, H, hd, dt=BF16X3)
    assert tuple(out.shape) == (NB * L, 2 * H * hd)
    q, k, v = (t.reshape(NB, L, H, hd).permute(0, 2, 1, 3) for t in qkv.double().reshape(NB * L, 3, H * hd).unbind(1))
    s = q @ k.transpose(-1, -2) / math.sqrt(hd)
    want = (torch.softmax(s, -1) @ v).permute(0, 2, 1, 3).reshape(NB * L, H * hd)
    err = (_unpair(out) - want).abs().max().item()
    assert err < 3e-5 * float(want.abs().max()), err
    want_lse = torch.logsumexp(s, -1) * math.log2(math.e)
    assert (lse.double() - want_lse).abs().max().item() < 1e-4


def _iou(result, teacher):
    m = O.IoUMeter()
    m.add_batch(result, teacher)
    return m.average()


@pytest.mark.parametrize("name", TRAINED_CASES)
def test_trained_fixture_bf16x3_mode_is_identical(name):
    """Same assertions as the float32 mode's: logits within 2e-4, loss within 1e-5, EVERY decision, fused cell and IoU digit."""
    c = Case(name)
    m = build(c, "bf16x3")
    m.eval()
    with torch.no_grad():
        loss, result, latent, ind = run(m, c)        # c.train fixtures: forward() (own-prediction fusion) without the autograd edge
    gp = c.tap("pred")
    torch.testing.assert_close(m._taps["pred"].cpu(), gp, rtol=2e-4, atol=2e-4)
    assert abs(loss.item() - c.loss) < 1e-5
    assert np.array_equal(ind.cpu().numpy(), c.ind)
    assert np.array_equal(result.cpu().numpy(), c.result)
    assert np.array_equal(_iou(result.cpu(), c.teacher), c.iou)


def test_bf16x3_refuses_training():
    c = Case(TRAINED_CASES[0])
    m = build(c, "bf16x3")
    m.train()
    c_train = c
    with pytest.raises(RuntimeError, match="inference mode"):
        bev, nxt = c.bev.to(DEV), c.next.to(DEV)
        m(bev, nxt, c.teacher.to(DEV), torch.from_numpy(c_train.trans), torch.from_numpy(c_train.num_agent).to(DEV), c.B, mask_ratio=c.mask_ratio)
