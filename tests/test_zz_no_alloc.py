"""The allocation-free entry point star_layernorm_bwd_static (include/star_hip.h) against star_layernorm_bwd, the
entry point it is a variant of: same kernels, the partial rows in the library's file-scope device array instead of a
caller workspace.  Called through the C ABI (ctypes), two calls in a row on the same GPU, the second with a smaller
input (fewer rows AND a narrower D, so stale partial rows of the first call lie behind what the second one reads).
Bar: bit-exact (integer/byte bar; the two paths run the same arithmetic in the same order)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from star_amd import _lib as L
    from star_amd import ops

DEV = "cuda"


def _case(m, d, dtype, seed, with_gin):
    g = torch.Generator(device="cpu").manual_seed(seed)
    x = torch.randn(m, d, generator=g).to(DEV)
    dy = torch.randn(m, d, generator=g).to(DEV).to(dtype)
    gamma = (1.0 + 0.1 * torch.randn(d, generator=g)).to(DEV)
    beta = (0.1 * torch.randn(d, generator=g)).to(DEV)
    g_in = torch.randn(m, d, generator=g).to(DEV) if with_gin else None
    dt = L.BF16 if dtype == torch.bfloat16 else L.F32
    _, mean, rstd = ops.layernorm_fwd(x, gamma, beta, 1e-6, dt)
    return x, dy, gamma, mean, rstd, g_in, dt


def _static(x, dy, gamma, mean, rstd, g_in, dt):
    m, d = x.shape
    g_out = torch.full((m, d), float("nan"), dtype=torch.float32, device=DEV)
    g_lp = torch.full((m, d), float("nan"), dtype=dy.dtype, device=DEV)
    dgamma = torch.full((d,), float("nan"), dtype=torch.float32, device=DEV)
    dbeta = torch.full((d,), float("nan"), dtype=torch.float32, device=DEV)
    L.check(L.lib().star_layernorm_bwd_static(L.p(dy), L.p(x), L.p(gamma), L.p(mean), L.p(rstd), L.p(g_in), L.p(g_out), L.p(g_lp),
                                              L.p(dgamma), L.p(dbeta), m, d, dt, L.stream()))
    return g_out, g_lp, dgamma, dbeta


def test_static_scratch_layernorm_bwd_matches_workspace_variant_twice():
    calls = [(40960 // 8 + 3, 384, torch.bfloat16, 1, True),     # more rows than partial-row blocks x waves: every block loops
             (257, 256, torch.float32, 2, False)]                # second call: smaller M, smaller D, other dtype, no g_in
    for m, d, dtype, seed, with_gin in calls:
        x, dy, gamma, mean, rstd, g_in, dt = _case(m, d, dtype, seed, with_gin)
        want = ops.layernorm_bwd(dy, x, gamma, mean, rstd, g_in=g_in)
        got = _static(x, dy, gamma, mean, rstd, g_in, dt)
        torch.cuda.synchronize()
        for w, g, name in zip(want, got, ("g_out", "g_lp", "dgamma", "dbeta")):
            assert torch.equal(w, g), (name, m, d)
        # and against float64 torch, so that "equal to the other entry point" is not "equally wrong"
        xd = x.double().requires_grad_(True)
        gd = gamma.double().requires_grad_(True)
        bd = torch.zeros(d, dtype=torch.float64, device=DEV, requires_grad=True)
        torch.nn.functional.layer_norm(xd, (d,), gd, bd, 1e-6).backward(dy.double())
        ref = xd.grad + (g_in.double() if g_in is not None else 0)
        tol = 2e-2 if dtype == torch.bfloat16 else 1e-4
        torch.testing.assert_close(got[0].double(), ref, rtol=tol, atol=tol)
        torch.testing.assert_close(got[2].double(), gd.grad, rtol=1e-3, atol=1e-2 if dtype == torch.bfloat16 else 1e-3)
        torch.testing.assert_close(got[3].double(), bd.grad, rtol=1e-3, atol=1e-2 if dtype == torch.bfloat16 else 1e-3)


def test_static_scratch_rejects_sizes_beyond_its_maximum():
    x, dy, gamma, mean, rstd, g_in, dt = _case(8, 2048, torch.float32, 3, False)
    out = torch.empty_like(x)
    rc = L.lib().star_layernorm_bwd_static(L.p(dy), L.p(x), L.p(gamma), L.p(mean), L.p(rstd), None, L.p(out), None, L.p(gamma.clone()),
                                           L.p(gamma.clone()), 8, 2048, dt, L.stream())
    assert rc != 0 and b"exceeds the built-in scratch" in L.lib().star_last_error()
