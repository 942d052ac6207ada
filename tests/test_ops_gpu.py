"""Per-kernel parity tests through the C ABI (run on the MI355X box: pytest -m gpu).

Integer / index / occupancy-map work is checked bit-exact against the oracle and the committed golden
vectors; floating-point kernels against the same op evaluated by PyTorch in float32/float64 with the
tolerance written next to each check."""
import math

import numpy as np
import pytest
import torch

from golden_util import GOLD, unpack

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from oracle import star_oracle as O
    from star_amd import ops
    from star_amd import _lib as L_mod
    from star_amd._lib import ACT_DGELU, ACT_DRELU, ACT_GELU, ACT_MULAUX, ACT_NONE, ACT_RELU, BF16, F32, GEMM_PRE_IS_GRAD

DEV = "cuda"


def _tol(dt):
    # bf16 storage: 8 significant bits in every stored operand/result; fp32 path: accumulation order only
    return dict(rtol=2e-2, atol=2e-2) if dt == 0 else dict(rtol=2e-5, atol=2e-5)


# ------------------------------------------------------------------------------------------------
def test_warp_fuse_golden_full_resolution_bit_exact():
    z = np.load(GOLD + "/warp_full256.npz")
    ind = unpack(z["ind_bits"], z["shape"], np.uint8)
    gt = unpack(z["gt_bits"], z["shape"])
    trans, na = z["trans"], z["num_agent"]
    B = trans.shape[0]
    pred = torch.from_numpy(ind).to(DEV)
    nad = torch.from_numpy(na).to(DEV)
    for tr, key in ((trans, "late_bits"), (trans.astype(np.float32), "late32_bits")):
        theta = ops.warp_thetas(torch.from_numpy(tr)).to(DEV)
        for binary in (False, True):   # generic uint8 path and the bit-packed path for {0,1} maps
            out = ops.warp_fuse(pred, theta, nad, B, binary=binary)
            assert np.array_equal(out.cpu().numpy(), unpack(z[key], z["shape"])), (key, binary)
    theta = ops.warp_thetas(torch.from_numpy(trans)).to(DEV)
    # ego = ground truth input, handed over as NHWC memory viewed NCHW (what CoModule does)
    ego = torch.from_numpy(gt).to(DEV).permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
    for binary in (False, True):
        out = ops.warp_fuse(pred, theta, nad.int(), B, ego=ego, binary=binary)
        assert np.array_equal(out.cpu().numpy(), unpack(z["ego_bits"], z["shape"])), binary


def test_warp_fuse_random_poses_vs_oracle():
    rng = np.random.RandomState(3)
    B, A, C, S = 2, 4, 13, 64
    ind = (rng.rand(A * B, C, S, S) < 0.1).astype(np.uint8)
    trans = np.zeros((B, A, A, 4, 4))
    for b in range(B):
        for i in range(A):
            for j in range(A):
                yaw, tx, ty = rng.uniform(-math.pi, math.pi), rng.uniform(-40, 40), rng.uniform(-40, 40)
                trans[b, i, j] = [[math.cos(yaw), -math.sin(yaw), 0, tx], [math.sin(yaw), math.cos(yaw), 0, ty], [0, 0, 1, 0], [0, 0, 0, 1]]
    na = np.array([[4] * A, [2] * A])
    want = O.late_fusion(ind.astype(np.float32), trans, na, B)
    for binary in (False, True):
        out = ops.warp_fuse(torch.from_numpy(ind).to(DEV), ops.warp_thetas(torch.from_numpy(trans)).to(DEV), torch.from_numpy(na).to(DEV), B,
                            binary=binary)
        assert np.array_equal(out.cpu().numpy(), want), binary


@pytest.mark.parametrize("R,L", [(5, 16), (7, 1024), (3, 342), (2, 4096), (40, 1024)])
def test_mask_sort_bit_exact(R, L):
    g = torch.Generator().manual_seed(R * 1000 + L)
    noise = torch.rand(R, L, generator=g)
    noise[0, : L // 2] = noise[0, L // 2: 2 * (L // 2)]     # ties: stable order must win
    s_ref, r_ref = O.shuffle_ids(noise)
    s, r = ops.mask_sort(noise.to(DEV))
    assert torch.equal(s.cpu(), s_ref) and torch.equal(r.cpu(), r_ref)


@pytest.mark.parametrize("T,random_mode,single", [(1, 0, 0), (2, 0, 0), (3, 0, 0), (3, 0, 1), (3, 1, 0), (3, 1, 1), (2, 1, 0)])
def test_rowmaps_reproduce_masking_and_unmasking(T, random_mode, single):
    """gather by src_row == the reference masking; scatter by dst_row (+fill) == the reference un-masking."""
    BA, L, D = 3, 16, 8
    g = torch.Generator().manual_seed(11 + T)
    x = torch.randn(BA * T, L, D, generator=g)
    ratio = 0.75
    if random_mode:
        noise = torch.rand(BA * T, L, generator=g)
        xm, mask_ref, ids_restore = O.random_masking(x, ratio, noise)
        keep = int(L * (1 - ratio))
    else:
        noise = torch.rand(BA, L, generator=g)
        xm, mask_ref, ids_restore = O.complement_masking(x, T, noise)
        keep = math.ceil(L * (1 - (1.0 - 1.0 / T)))
    ids_s, ids_r = ops.mask_sort(noise.to(DEV))
    assert torch.equal(ids_r.cpu(), ids_restore)
    src, dst, mask, fill = ops.build_rowmaps(ids_s, ids_r, BA, T, L, keep, random_mode, single)
    src, dst, mask, fill = src.cpu().long(), dst.cpu().long(), mask.cpu(), fill.cpu()
    assert torch.equal(mask, mask_ref)
    flat = torch.cat([x.reshape(-1, D), torch.zeros(1, D)])
    assert torch.equal(flat[src].reshape(BA * T, keep, D), xm)
    # un-masking: rows carry their timestamp's embedding; uncovered rows take the mask token
    temp = torch.randn(1, T, 1, D, generator=g)
    tok = torch.randn(1, 1, D, generator=g)
    if random_mode:
        want = O.random_unmasking(xm, mask_ref, ids_restore, T, temp, tok, False, bool(single)).permute(0, 2, 1)
    else:
        want = O.complement_unmasking(xm, ids_restore, T, temp, tok, False, bool(single))
    rows = xm.reshape(BA, T, keep, D) + temp
    got = torch.full((BA * L + 1, D), float("nan"))
    got[dst] = rows.reshape(-1, D)          # dst == -1 lands in the spare last row
    got = got[:-1]
    fillv = tok.reshape(D) + (temp[0, 0, 0] if random_mode else 0)
    got[fill.bool()] = fillv
    assert torch.equal(got.reshape(BA, L, D), want)


# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dt", [0, 1])
@pytest.mark.parametrize("M,D", [(80, 32), (1000, 384), (515, 256), (64, 768), (33, 1024)])
def test_layernorm_fwd_bwd(dt, M, D):
    g = torch.Generator().manual_seed(M + D)
    x = (torch.randn(M, D, generator=g) * 2 + 0.5).to(DEV)
    gamma = (1 + 0.1 * torch.randn(D, generator=g)).to(DEV)
    beta = (0.1 * torch.randn(D, generator=g)).to(DEV)
    y, mean, rstd = ops.layernorm_fwd(x, gamma, beta, 1e-6, dt)
    xr = x.double().requires_grad_(True)
    gr, br = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    yr = torch.nn.functional.layer_norm(xr, (D,), gr, br, 1e-6)
    torch.testing.assert_close(y.double(), yr.detach(), **_tol(dt))
    torch.testing.assert_close(mean.double(), xr.detach().mean(1), rtol=1e-5, atol=1e-5)
    dy = torch.randn(M, D, generator=g).to(DEV).to(y.dtype)
    g_in = torch.randn(M, D, generator=g).to(DEV)
    yr.backward(dy.double())
    g_out, g_lp, dgamma, dbeta = ops.layernorm_bwd(dy, x, gamma, mean, rstd, g_in=g_in)
    torch.testing.assert_close(g_out.double(), g_in.double() + xr.grad, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(g_lp.double(), g_out.double(), **_tol(dt))
    torch.testing.assert_close(dgamma.double(), gr.grad, rtol=1e-4, atol=1e-3)
    torch.testing.assert_close(dbeta.double(), br.grad, rtol=1e-4, atol=1e-3)


def _rand_lp(shape, g, dt, scale=1.0):
    t = (torch.randn(*shape, generator=g) * scale).to(DEV)
    return t.to(torch.bfloat16) if dt == 0 else t


@pytest.mark.parametrize("dt", [0, 1])
@pytest.mark.parametrize("M,N,K", [(80, 96, 32), (300, 384, 384), (1024, 1152, 384), (257, 32, 384), (130, 1664, 256), (64, 256, 1536), (100, 64, 832), (16, 48, 64)])
def test_gemm_plain(dt, M, N, K):
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    a, w = _rand_lp((M, K), g, dt), _rand_lp((N, K), g, dt, K ** -0.5)
    bias = torch.randn(N, generator=g).to(DEV)
    lp, f32, _ = ops.gemm(a, w, bias=bias, want_f32=True)
    ref = a.double() @ w.double().t() + bias.double()
    torch.testing.assert_close(f32.double(), ref, rtol=1e-4, atol=1e-4 if dt else 2e-3)
    torch.testing.assert_close(lp.double(), ref, **_tol(dt))


def test_gemm_asymmetric_identity():
    """A = I against an asymmetric W catches a transposed fragment or output map."""
    for dt in (0, 1):
        K = 64
        a = torch.eye(K, device=DEV).to(torch.bfloat16 if dt == 0 else torch.float32)
        w = (torch.arange(128 * K, device=DEV).reshape(128, K) % 61).float().to(a.dtype)
        _, f32, _ = ops.gemm(a, w, want_f32=True, want_lp=False)
        assert torch.equal(f32, w.float().t())


def test_gemm_res_ln_prefers_checks_alignment():
    """parameters that are views at a 4-byte offset into other storage must take the two-launch path, not raise in star_gemm_res_ln"""
    buf = torch.zeros(385, device=DEV)
    assert ops.gemm_res_ln_prefers(40960, 384, 384, 0)
    assert ops.gemm_res_ln_prefers(40960, 384, 384, 0, buf[:384])
    assert not ops.gemm_res_ln_prefers(40960, 384, 384, 0, buf[1:])
    from star_amd import engine
    torch.manual_seed(0)
    M, D = 40960, 384
    x = torch.randn(M, D, device=DEV)
    P = {}
    for name, shape in (("attn.qkv", (3 * D, D)), ("attn.proj", (D, D)), ("mlp.fc1", (4 * D, D)), ("mlp.fc2", (D, 4 * D))):
        P["b." + name + ".weight"] = torch.randn(*shape, device=DEV) * shape[1] ** -0.5
        P["b." + name + ".bias"] = torch.randn(shape[0], device=DEV) * 0.1
    for name in ("b.norm1", "b.norm2", "n"):
        P[name + ".weight"], P[name + ".bias"] = torch.ones(D, device=DEV), torch.zeros(D, device=DEV)
    ref, _ = engine._block_fwd(x.clone(), P, engine.Weights(), "b.", 12, 40, 1024, 0, 1e-6, None, False, nxt="n")
    odd = torch.zeros(D + 1, device=DEV)
    odd[1:] = P["b.attn.proj.bias"]
    P["b.attn.proj.bias"] = odd[1:]                                 # same values, 4-byte offset
    out, _ = engine._block_fwd(x.clone(), P, engine.Weights(), "b.", 12, 40, 1024, 0, 1e-6, None, False, nxt="n")
    torch.testing.assert_close(out, ref, rtol=1e-2, atol=2e-2)       # the two paths differ by bf16 rounding points only


@pytest.mark.parametrize("geo", [None, "0", "1", "2", "3"])
@pytest.mark.parametrize("dt", [0, 1])
def test_gemm_epilogues(dt, geo, monkeypatch):
    """geo: STAR_AMD_GEMM_GEO forces one tile geometry of the bf16 fast kernel (read per call), so that every specialised epilogue
    is exercised on every geometry at unit level and not only on the one the size heuristic picks for these small shapes"""
    if geo is not None:
        if dt == 1:
            pytest.skip("tile geometries belong to the bf16 kernel")
        monkeypatch.setenv("STAR_AMD_GEMM_GEO", geo)
    g = torch.Generator().manual_seed(5)
    M, N, K, L = 200, 128, 64, 50
    a, w = _rand_lp((M, K), g, dt), _rand_lp((N, K), g, dt, K ** -0.5)
    bias = torch.randn(N, generator=g).to(DEV)
    base = a.double() @ w.double().t() + bias.double()
    # GELU with pre-activation copy
    lp, _, pre = ops.gemm(a, w, bias=bias, act=ACT_GELU, want_pre=True)
    torch.testing.assert_close(pre.double(), base, **_tol(dt))
    torch.testing.assert_close(lp.double(), torch.nn.functional.gelu(base), **_tol(dt))
    # ReLU, float32 + low-precision outputs
    lp, f32, _ = ops.gemm(a, w, bias=bias, act=ACT_RELU, want_f32=True)
    torch.testing.assert_close(f32.double(), base.clamp_min(0), rtol=1e-4, atol=2e-3)
    # residual in place (x += A W^T + b)
    x = torch.randn(M, N, generator=g).to(DEV)
    x0 = x.clone()
    ops.gemm(a, w, bias=bias, res=x, out_f32=x, want_lp=False)
    torch.testing.assert_close(x.double(), x0.double() + base, rtol=1e-4, atol=2e-3)
    # periodic table (pos embed) + per-timestamp table (temp embed)
    pos = torch.randn(L, N, generator=g).to(DEV)
    temp = torch.randn(4, N, generator=g).to(DEV)
    _, f32, _ = ops.gemm(a, w, bias=bias, res=pos, res_mod=L, res2=temp, res2_div=25, res2_mod=4, want_f32=True, want_lp=False)
    rows = torch.arange(M, device=DEV)
    ref = base + pos.double()[rows % L] + temp.double()[(rows // 25) % 4]
    torch.testing.assert_close(f32.double(), ref, rtol=1e-4, atol=2e-3)
    # gathered A rows (-1 = zero row) and scattered C rows (-1 = dropped)
    a_rows = torch.randint(-1, M, (150,), generator=g).int().to(DEV)
    perm = torch.randperm(180, generator=g)[:150].int()
    perm[::7] = -1
    c_rows = perm.to(DEV)
    _, f32, _ = ops.gemm(a, w, bias=bias, a_rows=a_rows, c_rows=c_rows, out_rows=180, res=pos, res_mod=L, want_f32=True, want_lp=False)
    az = torch.cat([a.double(), torch.zeros(1, K, device=DEV, dtype=torch.double)])
    full = az[a_rows.long()] @ w.double().t() + bias.double()
    ref = torch.zeros(180, N, device=DEV, dtype=torch.double)
    ok = c_rows >= 0
    dest = c_rows[ok].long()
    ref[dest] = full[ok] + pos.double()[dest % L]
    torch.testing.assert_close(f32.double(), ref, rtol=1e-4, atol=2e-3)
    # activation gradients
    u = _rand_lp((M, N), g, dt)
    lp, _, _ = ops.gemm(a, w, act=ACT_DGELU, aux=u)
    ud = u.double().requires_grad_(True)
    torch.nn.functional.gelu(ud).sum().backward()
    torch.testing.assert_close(lp.double(), (a.double() @ w.double().t()) * ud.grad, **_tol(dt))
    lp, _, _ = ops.gemm(a, w, act=ACT_DRELU, aux=u)
    torch.testing.assert_close(lp.double(), (a.double() @ w.double().t()) * (u.double() > 0), **_tol(dt))
    # the forward can store act'(pre-activation) instead (PRE_IS_GRAD); the backward then only multiplies (MULAUX)
    lp, _, dpre = ops.gemm(a, w, bias=bias, act=ACT_GELU, want_pre=True, flags=GEMM_PRE_IS_GRAD)
    bd = base.clone().requires_grad_(True)
    torch.nn.functional.gelu(bd).sum().backward()
    torch.testing.assert_close(lp.double(), torch.nn.functional.gelu(base), **_tol(dt))
    torch.testing.assert_close(dpre.double(), bd.grad, **_tol(dt))
    lp, _, dpre = ops.gemm(a, w, bias=bias, act=ACT_RELU, want_pre=True, flags=GEMM_PRE_IS_GRAD)
    clear = base.abs() > 0.05
    assert torch.equal(dpre.double()[clear], (base > 0).double()[clear])
    lp, _, _ = ops.gemm(a, w, act=ACT_MULAUX, aux=u)
    torch.testing.assert_close(lp.double(), (a.double() @ w.double().t()) * u.double(), **_tol(dt))


@pytest.mark.parametrize("dt", [0, 1])
@pytest.mark.parametrize("M,N,K", [(80, 96, 32), (5000, 384, 1152), (1024, 1536, 384), (700, 32, 384), (333, 1664, 256), (40960, 384, 384)])
def test_gemm_wgrad(dt, M, N, K):
    g = torch.Generator().manual_seed(M + N + K)
    dy, x = _rand_lp((M, N), g, dt), _rand_lp((M, K), g, dt)
    dw, db = ops.gemm_wgrad(dy, x)
    ref = dy.double().t() @ x.double()
    scale = math.sqrt(M)
    torch.testing.assert_close(dw.double() / scale, ref / scale, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(db.double() / scale, dy.double().sum(0) / scale, rtol=1e-4, atol=1e-4)
    # gathered X rows
    rows = torch.randint(-1, M, (M,), generator=g).int().to(DEV)
    dw2, _ = ops.gemm_wgrad(dy, x, x_rows=rows, want_bias=False)
    xz = torch.cat([x.double(), torch.zeros(1, K, device=DEV, dtype=torch.double)])
    torch.testing.assert_close(dw2.double() / scale, (dy.double().t() @ xz[rows.long()]) / scale, rtol=1e-4, atol=1e-4)


def _wg_group_jobs(shapes, g, scale=1.0):
    jobs = []
    for M, N, K in shapes:
        dy = (_rand_lp((M, N), g, 0).float() * scale).to(torch.bfloat16)
        jobs.append((dy, _rand_lp((M, K), g, 0), None, None))
    return jobs


@pytest.mark.parametrize("shapes", [
    [(40960, 1152, 384), (40960, 384, 384), (40960, 1536, 384), (40960, 384, 1536)],      # an encoder block: 108 tiles, split 2
    [(10240, 768, 256), (10240, 256, 256), (10240, 1024, 256), (10240, 256, 1024)],        # a decoder block, fewer rows: split 5
    [(5000, 1664, 256), (333, 96, 32), (4097, 384, 832)],                                  # ragged rows, partial tiles, unequal M
    [(300, 128, 128)],                                                                     # one tile, one split: no fold at all
])
def test_gemm_wgrad_group(shapes):
    """star_gemm_wgrad_group against float64, against star_gemm_wgrad, and bit for bit against itself over repeated launches that
    reuse one workspace with DIFFERENT data in between (a stale partial tile or ticket would show) while another stream keeps
    the chip busy (uneven load: the fold's cross-workgroup hand-off must not depend on an idle chip)."""
    g = torch.Generator().manual_seed(len(shapes) * 7 + shapes[0][0])
    jobs = _wg_group_jobs(shapes, g)
    other = _wg_group_jobs(shapes, g, scale=3.0)
    first = [(dw.clone(), db.clone()) for dw, db in ops.gemm_wgrad_group(jobs)]
    for (dy, x, _, _), (dw, db) in zip(jobs, first):
        scale = math.sqrt(dy.shape[0])
        torch.testing.assert_close(dw.double() / scale, (dy.double().t() @ x.double()) / scale, rtol=1e-4, atol=1e-4)
        torch.testing.assert_close(db.double() / scale, dy.double().sum(0) / scale, rtol=1e-4, atol=1e-4)
        dw1, db1 = ops.gemm_wgrad(dy, x)
        torch.testing.assert_close(dw, dw1, rtol=1e-5, atol=1e-3 * scale / 64)
        torch.testing.assert_close(db, db1, rtol=1e-5, atol=1e-3 * scale / 64)
    side = torch.cuda.Stream()
    big = torch.randn(64 << 20, device=DEV)
    for it in range(6):
        with torch.cuda.stream(side):          # background traffic on every XCD
            for _ in range(4):
                big.mul_(1.0001)
        ops.gemm_wgrad_group(other)
        again = ops.gemm_wgrad_group(jobs)
        for (dw, db), (dw0, db0) in zip(again, first):
            assert torch.equal(dw, dw0) and torch.equal(db, db0), "grouped weight gradient not reproducible (launch %d)" % it
    torch.cuda.synchronize()
    # no bias requested for one job; caller-provided outputs
    dy, x = jobs[0][0], jobs[0][1]
    dw = torch.full((dy.shape[1], x.shape[1]), float("nan"), device=DEV)
    (dw_o, db_o), = ops.gemm_wgrad_group([(dy, x, dw, False)])
    assert db_o is None and dw_o.data_ptr() == dw.data_ptr()
    # tile shape and split count are functions of the job alone: launched on its own (or in any other company) a product comes out
    # with the SAME bits as inside the group
    assert torch.equal(dw, first[0][0])
    if len(jobs) > 2:
        regrouped = ops.gemm_wgrad_group(jobs[2:] + jobs[:1]) + ops.gemm_wgrad_group(jobs[1:2])
        for (dw_r, db_r), k in zip(regrouped, list(range(2, len(jobs))) + [0, 1]):
            assert torch.equal(dw_r, first[k][0]) and torch.equal(db_r, first[k][1])


@pytest.mark.parametrize("n,h,w,cin,cout", [(3, 16, 32, 128, 32), (2, 32, 32, 64, 128), (20, 8, 8, 256, 48), (1, 64, 128, 32, 256)])
def test_gemm_conv3x3(n, h, w, cin, cout):
    """star_gemm's implicit 3 x 3 convolution (conv word: the nine taps as k-blocks of one product) against torch's conv2d in float64
    on the same bf16-rounded operands; with and without bias, float32 and bf16 outputs."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(h * 7 + cin)
    x = torch.randn(n, h, w, cin, generator=g).to(DEV).bfloat16()
    wt = (torch.randn(cout, cin, 3, 3, generator=g) * (9 * cin) ** -0.5).to(DEV).bfloat16()
    bias = torch.randn(cout, generator=g).to(DEV)
    wf = wt.permute(0, 2, 3, 1).reshape(cout, 9 * cin).contiguous()          # [N, (ky, kx, c)]
    want = F.conv2d(x.double().permute(0, 3, 1, 2), wt.double(), bias.double(), padding=1).permute(0, 2, 3, 1).reshape(n * h * w, cout)
    lp, f32, _ = ops.gemm(x.reshape(n * h * w, cin), wf, bias=bias, want_lp=True, want_f32=True, conv=(h, w))
    torch.testing.assert_close(f32.double(), want, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(lp.double(), want, rtol=1e-2, atol=1e-2)
    _, f32b, _ = ops.gemm(x.reshape(n * h * w, cin), wf, want_lp=False, want_f32=True, conv=(h, w))
    torch.testing.assert_close(f32b.double(), want - bias.double(), rtol=1e-4, atol=1e-4)
    with pytest.raises(RuntimeError):       # rows that are not whole images
        ops.gemm(x.reshape(n * h * w, cin)[: n * h * w - 16], wf, want_lp=False, want_f32=True, conv=(h, w))


@pytest.mark.parametrize("n,h,w,N,K", [(3, 16, 32, 32, 128), (2, 32, 32, 128, 256), (5, 8, 16, 256, 512), (2, 64, 64, 8, 64)])
def test_gemm_wgrad_group_conv_taps(n, h, w, N, K):
    """the nine taps of a 3 x 3 convolution's weight gradient as grouped jobs (tap mode: the shifted-pixel gather is arithmetic in
    the kernel) against float64 over the explicit shifted-row table, and against star_gemm_wgrad's x_rows path."""
    from star_amd.joint import conv_rows
    g = torch.Generator().manual_seed(n * 100 + K)
    M = n * h * w
    dy = torch.randn(M, N, generator=g).to(DEV).bfloat16()
    x = torch.randn(M, K, generator=g).to(DEV).bfloat16()
    rows = conv_rows(n, h, w, DEV)
    out = ops.gemm_wgrad_group([(dy, x, None, None if t == 0 else False, ops.wgrad_tap(t // 3 - 1, t % 3 - 1, h, w)) for t in range(9)])
    scale = math.sqrt(M)
    for t in range(9):
        r = rows[t].long()
        xs = torch.where((r >= 0).unsqueeze(1), x[r.clamp(min=0)].double(), torch.zeros((), dtype=torch.float64, device=DEV))
        torch.testing.assert_close(out[t][0].double() / scale, (dy.double().t() @ xs) / scale, rtol=1e-4, atol=1e-4)
        dw1, _ = ops.gemm_wgrad(dy, x, x_rows=rows[t], want_bias=False)
        torch.testing.assert_close(out[t][0], dw1, rtol=1e-5, atol=1e-3 * scale / 64)
    torch.testing.assert_close(out[0][1].double() / scale, dy.double().sum(0) / scale, rtol=1e-4, atol=1e-4)
    assert all(o[1] is None for o in out[1:])
    again = ops.gemm_wgrad_group([(dy, x, None, False, ops.wgrad_tap(t // 3 - 1, t % 3 - 1, h, w)) for t in (4, 0, 8)])
    assert torch.equal(again[0][0], out[4][0]) and torch.equal(again[1][0], out[0][0]) and torch.equal(again[2][0], out[8][0])
    with pytest.raises(RuntimeError):       # rows that are not whole images
        ops.gemm_wgrad_group([(dy[: M - 32], x[: M - 32], None, False, ops.wgrad_tap(0, 0, h, w))])


@pytest.mark.parametrize("m,n,k,bias", [(40960, 256, 256, False), (40960 - 37, 256, 1024, True), (81920, 256, 768, False), (40960, 384, 1152, False)])
def test_gemm_rows_path(m, n, k, bias, monkeypatch):
    """plain narrow Linears at large M go through the whole-row kernel (gemm_rows_kernel, reached inside star_gemm): against float64 and
    against the tiled kernel (STAR_AMD_GEMM_ROWS is read once per process, so the tiled result comes from a wider problem's columns)."""
    g = torch.Generator().manual_seed(m + k)
    a = torch.randn(m, k, generator=g).to(DEV).bfloat16()
    w = (torch.randn(n, k, generator=g) * k ** -0.5).to(DEV).bfloat16()
    b = torch.randn(n, generator=g).to(DEV) if bias else None
    out, _, _ = ops.gemm(a, w, bias=b)
    ref = a[:4096].double() @ w.double().t() + (b.double() if bias else 0.0)
    torch.testing.assert_close(out[:4096].double(), ref, rtol=1e-2, atol=1e-2)
    ref_t = a[-512:].double() @ w.double().t() + (b.double() if bias else 0.0)
    torch.testing.assert_close(out[-512:].double(), ref_t, rtol=1e-2, atol=1e-2)
    # the tiled kernel on the same rows (a float32 output keeps it off the whole-row path): same products, float32 accumulation both ways
    _, o32, _ = ops.gemm(a, w, bias=b, want_lp=False, want_f32=True)
    if bias:        # the whole-row kernel starts its accumulators from the bias, the tiled one adds it last: one float32 rounding apart
        torch.testing.assert_close(out.float(), o32, rtol=1e-2, atol=1e-2)
    else:
        assert torch.equal(out, o32.bfloat16())


def test_gemm_wgrad_group_rejects():
    dy, x = torch.zeros(64, 16, device=DEV), torch.zeros(64, 16, device=DEV)      # float32 operands
    with pytest.raises((RuntimeError, AssertionError)):
        ops.gemm_wgrad_group([(dy, x, None, None)])
    assert not ops.wgrad_group_eligible(dy, x)
    b = torch.zeros(64, 16, device=DEV, dtype=torch.bfloat16)
    assert ops.wgrad_group_eligible(b, b) and not ops.wgrad_group_eligible(b, b, x_rows=torch.zeros(64, dtype=torch.int32, device=DEV))


def test_gemm_wgrad_asymmetric():
    M, N, K = 64, 16, 16
    dy = torch.zeros(M, N, device=DEV, dtype=torch.bfloat16)
    x = torch.zeros(M, K, device=DEV, dtype=torch.bfloat16)
    dy[5, 3] = 1.0
    x[5, 9] = 2.0
    dw, _ = ops.gemm_wgrad(dy, x)
    want = torch.zeros(N, K, device=DEV)
    want[3, 9] = 2.0
    assert torch.equal(dw, want)


# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dt", [0, 1])
@pytest.mark.parametrize("channels_last", [False, True])
def test_patchify(dt, channels_last):
    BA, T, C, S, p = 3, 3, 13, 32, 8
    g = torch.Generator().manual_seed(1)
    img1 = (torch.rand(BA, C, S, S, generator=g) < 0.3).float().to(DEV)
    nxt = (torch.rand(BA, T - 1, C, S, S, generator=g) < 0.3).float().to(DEV)
    frames = torch.cat([img1.unsqueeze(1), nxt], 1).reshape(BA * T, C, S, S)
    gh = S // p
    pat = frames.reshape(BA * T, C, gh, p, gh, p)
    if channels_last:
        img1 = img1.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
        nxt = nxt.permute(0, 1, 3, 4, 2).contiguous().permute(0, 1, 4, 2, 3)
        want = pat.permute(0, 2, 4, 3, 5, 1)      # (py, px, c)
    else:
        want = pat.permute(0, 2, 4, 1, 3, 5)      # (c, py, px)
    out = ops.patchify(img1, nxt, T, p, dt, channels_last_order=channels_last)
    assert torch.equal(out.float(), want.reshape(BA * T * gh * gh, -1))


@pytest.mark.parametrize("dt", [0, 1])
def test_ce_argmax(dt):
    BA, C, S, p = 3, 13, 32, 8
    g = torch.Generator().manual_seed(2)
    pred = (torch.randn(BA, 2, C, S, S, generator=g) * 2).to(DEV)
    pred[0, 1, 0, 0, :8] = pred[0, 0, 0, 0, :8]          # exact ties -> class 0
    target = (torch.rand(BA, C, S, S, generator=g) < 0.3).float().to(DEV)
    target_cl = target.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
    # patch layout: column = cls*V + (c*p + py)*p + px
    gh = S // p
    logits = pred.reshape(BA, 2, C, gh, p, gh, p).permute(0, 3, 5, 1, 2, 4, 6).reshape(BA * gh * gh, -1).contiguous()
    r = ops.ce_argmax(logits, target_cl, p, dt, want_pred=True)
    pd = pred.double().requires_grad_(True)
    loss = torch.nn.functional.cross_entropy(pd, target.long())
    loss.backward()
    assert abs(r["loss"].item() - loss.item()) < 1e-5
    want_ind = torch.argmax(torch.softmax(pred, 1), 1)
    assert torch.equal(r["ind64"], want_ind) and torch.equal(r["ind8"].long(), want_ind)
    assert torch.equal(r["pred"], pred)
    dref = pd.grad.reshape(BA, 2, C, gh, p, gh, p).permute(0, 3, 5, 1, 2, 4, 6).reshape(BA * gh * gh, -1)
    n = BA * C * S * S
    torch.testing.assert_close(r["dlogits"].double() * n, dref * n, rtol=1e-2 if dt == 0 else 1e-5, atol=4e-3 if dt == 0 else 1e-6)


def test_transpose_and_fill_rows():
    BA, L, D = 3, 16, 32
    g = torch.Generator().manual_seed(4)
    src = torch.randn(BA * L, D, generator=g).to(DEV)
    pos = torch.randn(L, D, generator=g).to(DEV)
    out = ops.transpose_rows(src, pos, BA, L, D)
    want = src.reshape(BA, L, D).permute(0, 2, 1).reshape(BA, L, D) + pos
    assert torch.equal(out.reshape(BA, L, D), want)
    gd = torch.randn(BA * L, D, generator=g).to(DEV)
    gs = ops.transpose_rows_bwd(gd, BA, L, D)
    assert torch.equal(gs.reshape(BA, L, D), gd.reshape(BA, D, L).permute(0, 2, 1))
    fill = (torch.rand(BA * L, generator=g) < 0.4).int().to(DEV)
    tok, t0 = torch.randn(D, generator=g).to(DEV), torch.randn(D, generator=g).to(DEV)
    x = src.clone()
    ops.fill_rows(x, fill, tok, t0, pos, BA, L)
    want = torch.where(fill.bool().unsqueeze(1), (tok + t0).unsqueeze(0) + pos.repeat(BA, 1), src)
    assert torch.equal(x, want)


def test_cast_adamw_sumsq():
    g = torch.Generator().manual_seed(6)
    n = 100003
    x = torch.randn(n, generator=g).to(DEV)
    assert torch.equal(ops.cast(x, 0), x.to(torch.bfloat16))
    assert torch.equal(ops.cast_back(ops.cast(x, 0)), x.to(torch.bfloat16).float())
    assert abs(ops.sumsq(x).item() / (x.double() ** 2).sum().item() - 1) < 1e-6
    pr = torch.nn.Parameter(torch.randn(n, generator=g).to(DEV))
    opt = torch.optim.AdamW([pr], lr=1e-3, betas=(0.9, 0.95), weight_decay=0.05)
    p2 = pr.detach().clone()
    m, v = torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    for step in range(1, 4):
        grad = torch.randn(n, generator=g).to(DEV)
        pr.grad = grad.clone()
        opt.step()
        ops.adamw(p2, grad, m, v, None, 1e-3, 0.9, 0.95, 1e-8, 0.05, step)
        torch.testing.assert_close(p2, pr.detach(), rtol=1e-5, atol=1e-6)


def test_iou_confusion_vs_oracle():
    z = np.load(GOLD + "/standalone.npz")
    pred = unpack(z["iou_pred_bits"], z["iou_shape"])
    tgt = unpack(z["iou_tgt_bits"], z["iou_shape"])
    for b in range(pred.shape[0]):
        conf = ops.iou_confusion(torch.from_numpy(pred[b]).to(DEV), torch.from_numpy(tgt[b]).to(DEV)).cpu().numpy()
        want = O.occupancy_confusions(torch.from_numpy(pred[b]), torch.from_numpy(tgt[b]))
        assert np.array_equal(conf, want)
        assert np.array_equal(np.array([O.occupancy_iou(c) for c in conf]), z["iou_every_batch"][b])


@pytest.mark.parametrize("dt", [0, 1])
@pytest.mark.parametrize("m,n,k", [(256, 128, 384), (70, 32, 96)])        # the fast kernel's shape and a generic-kernel shape
def test_relu_epilogue_keeps_nan_like_torch_relu(dt, m, n, k):
    """torch.relu(NaN) is NaN: the reference's non-finite-loss stop (CoModule.py:116-122) depends on a NaN reaching the loss through
    the compressor's ReLU.  (A max-based ReLU returns 0 for NaN and would train on.)"""
    g = torch.Generator().manual_seed(3)
    a = _rand_lp((m, k), g, dt)
    w = _rand_lp((n, k), g, dt)
    a[5, 7] = float("nan")
    lp, f32, _ = ops.gemm(a, w, act=ACT_RELU, want_f32=True)
    want = torch.relu(a.double() @ w.double().t())
    assert torch.isnan(want[5]).all() and torch.isnan(f32[5]).all() and torch.isnan(lp[5]).all()
    keep = torch.ones(m, dtype=torch.bool, device=DEV)
    keep[5] = False
    assert not torch.isnan(f32[keep]).any()
    torch.testing.assert_close(f32[keep].double(), want[keep], rtol=2e-2 if dt == 0 else 1e-4, atol=2e-2 if dt == 0 else 1e-4)


# ------------------------------------------------------------------------------------------------
def _attn_ref(qkv, NB, L, H, hd, dout=None):
    x = qkv.double().reshape(NB, L, 3, H, hd).permute(2, 0, 3, 1, 4).detach().requires_grad_(True)
    q, k, v = x[0], x[1], x[2]
    att = ((q @ k.transpose(-2, -1)) * hd ** -0.5).softmax(-1)
    out = (att @ v).transpose(1, 2).reshape(NB * L, H * hd)
    if dout is None:
        return out.detach(), None
    out.backward(dout.double())
    return out.detach(), x.grad.permute(1, 3, 0, 2, 4).reshape(NB * L, 3 * H * hd)


@pytest.mark.parametrize("dt", [0, 1])
@pytest.mark.parametrize("NB,L,H,hd", [(2, 16, 2, 16), (3, 64, 3, 32), (2, 342, 2, 16), (1, 342, 3, 32), (2, 1024, 2, 32),
                                       (1, 1024, 4, 16), (2, 256, 2, 64), (1, 100, 1, 64), (1, 4096, 1, 32)])
def test_attention_fwd_bwd(dt, NB, L, H, hd):
    g = torch.Generator().manual_seed(NB + L + H + hd)
    qkv = _rand_lp((NB * L, 3 * H * hd), g, dt, 1.5)
    dout = _rand_lp((NB * L, H * hd), g, dt)
    out, lse = ops.attn_fwd(qkv, NB, L, H, hd)
    ref, dref = _attn_ref(qkv, NB, L, H, hd, dout)
    tol = dict(rtol=2e-2, atol=1e-2) if dt == 0 else dict(rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(out.double(), ref, **tol)
    # log2-domain LSE against the definition
    x = qkv.double().reshape(NB, L, 3, H, hd)
    s = torch.einsum("blhd,bmhd->bhlm", x[:, :, 0], x[:, :, 1]) * hd ** -0.5
    torch.testing.assert_close(lse.double(), torch.logsumexp(s, -1) / math.log(2.0), rtol=1e-3 if dt == 0 else 1e-5, atol=1e-3 if dt == 0 else 1e-5)
    dqkv = ops.attn_bwd(qkv, out, dout, lse, NB, L, H, hd)
    tolb = dict(rtol=3e-2, atol=4e-2) if dt == 0 else dict(rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(dqkv.double(), dref, **tolb)


def test_attention_one_hot_rows_catch_layout_errors():
    """Integer data: V rows one-hot, q.k chosen so that every query attends (almost) only to one key."""
    NB, L, H, hd = 1, 64, 1, 32
    qkv = torch.zeros(NB * L, 3, H, hd, device=DEV)
    tgt = (torch.arange(L, device=DEV) * 7 + 3) % L
    code = lambda idx: torch.stack([((idx >> b) & 1).float() * 2 - 1 for b in range(6)], 1) * 16.0  # noqa: E731
    qkv[:, 0, 0, :6] = code(tgt)
    qkv[:, 1, 0, :6] = code(torch.arange(L, device=DEV))
    qkv[:, 2, 0, :] = (torch.arange(L, device=DEV).unsqueeze(1) * 5 + torch.arange(hd, device=DEV).unsqueeze(0)) % 11
    q16 = qkv.reshape(NB * L, -1).to(torch.bfloat16)
    out, _ = ops.attn_fwd(q16, NB, L, H, hd)
    want = qkv[tgt, 2, 0, :]
    torch.testing.assert_close(out.float(), want, rtol=0, atol=1e-2)


def test_cast_batch_matches_single_casts():
    g = torch.Generator().manual_seed(11)
    for dt, td in ((BF16, torch.bfloat16), (F32, torch.float32)):
        jobs = []
        for (r, c, tr) in [(384, 384, False), (1152, 384, True), (70, 130, True), (33, 64, False), (1, 8, True)]:
            src = torch.randn(r, c, generator=g).to(DEV)
            dst = torch.empty((c, r) if tr else (r, c), dtype=td, device=DEV)
            jobs.append((src, dst, tr))
        tab, n, tiles = ops.cast_batch_table(jobs, DEV)
        ops.cast_batch(tab, n, tiles, dt)
        for src, dst, tr in jobs:
            want = ops.cast_transpose(src, dt) if tr else ops.cast(src, dt)
            assert torch.equal(dst, want), (src.shape, tr)
        # row-gathered jobs that share one destination (the fused prediction-head weight: two Linear weights, class-major rows), plain
        # and transposed, plus a float32 job (its bias) in the same launch
        v, k = 832, 256
        a, b = torch.randn(v, k, generator=g).to(DEV), torch.randn(v, k, generator=g).to(DEV)
        ba, bb = torch.randn(v, generator=g).to(DEV), torch.randn(v, generator=g).to(DEV)
        idx = torch.randperm(v, generator=g).to(torch.int32).to(DEV)
        plain = torch.full((2 * v, k), 7.0, dtype=td, device=DEV)
        trans = torch.full((k, 2 * v), 7.0, dtype=td, device=DEV)
        bias = torch.full((2 * v,), 7.0, dtype=torch.float32, device=DEV)
        jobs = [(a, plain, False, dict(src_rows=idx, dst_ld=k, dst_off=0)), (b, plain, False, dict(src_rows=idx, dst_ld=k, dst_off=v * k)),
                (a, trans, True, dict(src_rows=idx, dst_ld=2 * v, dst_off=0)), (b, trans, True, dict(src_rows=idx, dst_ld=2 * v, dst_off=v)),
                (ba.reshape(v, 1), bias, False, dict(src_rows=idx, dst_ld=1, dst_off=0, dst_f32=True)),
                (bb.reshape(v, 1), bias, False, dict(src_rows=idx, dst_ld=1, dst_off=v, dst_f32=True))]
        tab, n, tiles = ops.cast_batch_table(jobs, DEV)
        ops.cast_batch(tab, n, tiles, dt)
        want = torch.cat([a[idx.long()], b[idx.long()]], 0)
        assert torch.equal(plain, ops.cast(want, dt)) and torch.equal(trans, ops.cast_transpose(want, dt))
        assert torch.equal(bias, torch.cat([ba[idx.long()], bb[idx.long()]]))
    with pytest.raises(AssertionError, match="past its destination"):
        ops.cast_batch_table([(torch.zeros(8, 8, device=DEV), torch.zeros(8, 8, device=DEV), False, dict(dst_ld=8, dst_off=1))], DEV)


def test_warp_fuse_full_size_vs_c_oracle():
    """BASELINE-size maps (256 x 256 x 13, 5 agents, a scene with padded slots) against the plain-C restatement."""
    from oracle import c_oracle as OC
    rng = np.random.RandomState(9)
    B, A, C, S = 2, 5, 13, 256
    ind = (rng.rand(A * B, C, S, S) < 0.05).astype(np.uint8)
    trans = np.zeros((B, A, A, 4, 4))
    for b in range(B):
        for i in range(A):
            for j in range(A):
                yaw, tx, ty = rng.uniform(-math.pi, math.pi), rng.uniform(-20, 20), rng.uniform(-20, 20)
                trans[b, i, j] = [[math.cos(yaw), -math.sin(yaw), 0, tx], [math.sin(yaw), math.cos(yaw), 0, ty], [0, 0, 1, 0], [0, 0, 0, 1]]
    na = np.array([[5] * A, [3] * A])
    want = OC.late_fusion(ind.astype(np.float32), trans, na, B)
    for binary in (False, True):
        out = ops.warp_fuse(torch.from_numpy(ind).to(DEV), ops.warp_thetas(torch.from_numpy(trans)).to(DEV), torch.from_numpy(na).to(DEV), B,
                            binary=binary)
        assert np.array_equal(out.cpu().numpy(), want), binary
    ego = (rng.rand(A * B, C, S, S) < 0.05).astype(np.float32)
    want = OC.late_fusion(ind.astype(np.float32), trans, na, B, ego_maps=ego)
    out = ops.warp_fuse(torch.from_numpy(ind).to(DEV), ops.warp_thetas(torch.from_numpy(trans)).to(DEV), torch.from_numpy(na).to(DEV), B,
                        ego=torch.from_numpy(ego).to(DEV), binary=True)
    assert np.array_equal(out.cpu().numpy(), want)


def test_warp_fuse_edge_cases_vs_oracle():
    """num_agent 0 (nothing fused), 1 (only the ego), a count larger than A (clamped as range() over a too-short list would
    not be: the oracle is given A); identity poses (every neighbour lands on itself); translations far outside the map
    (every sample padded with zeros); non-finite pose entries (comparisons fail -> zero padding, like grid_sample)."""
    from oracle import c_oracle as OC
    rng = np.random.RandomState(17)
    B, A, C, S = 4, 3, 13, 32
    ind = (rng.rand(A * B, C, S, S) < 0.2).astype(np.uint8)
    trans = np.zeros((B, A, A, 4, 4))
    for b in range(B):
        for i in range(A):
            for j in range(A):
                trans[b, i, j] = np.eye(4)
    trans[1, :, :, 0, 3] = 1.0e4        # scene 1: far outside
    trans[1, :, :, 1, 3] = -3.0e3
    for i in range(A):                 # scene 3: generic poses
        for j in range(A):
            yaw = rng.uniform(-math.pi, math.pi)
            trans[3, i, j] = [[math.cos(yaw), -math.sin(yaw), 0, rng.uniform(-30, 30)], [math.sin(yaw), math.cos(yaw), 0, rng.uniform(-30, 30)],
                              [0, 0, 1, 0], [0, 0, 0, 1]]
    for counts in ([0, 1, 3, 3], [3, 3, 1, 0], [1, 0, 0, 2]):
        na = np.repeat(np.array(counts)[:, None], A, axis=1)
        want = OC.late_fusion(ind.astype(np.float32), trans, na, B)
        assert np.array_equal(want, O.late_fusion(ind.astype(np.float32), trans, na, B))
        for binary in (False, True):
            out = ops.warp_fuse(torch.from_numpy(ind).to(DEV), ops.warp_thetas(torch.from_numpy(trans)).to(DEV), torch.from_numpy(na).to(DEV), B,
                                binary=binary)
            assert np.array_equal(out.cpu().numpy(), want), (counts, binary)
    # identity poses with all agents: every ego map is the union of the scene's maps
    na = np.full((B, A), A)
    out = ops.warp_fuse(torch.from_numpy(ind).to(DEV), ops.warp_thetas(torch.from_numpy(trans)).to(DEV), torch.from_numpy(na).to(DEV), B, binary=True)
    x = ind.reshape(A, B, C, S, S)
    assert np.array_equal(out.cpu().numpy().reshape(A, B, C, S, S)[:, 0], np.broadcast_to(x[:, 0].max(0), (A, C, S, S)).astype(np.float32))
    assert np.array_equal(out.cpu().numpy().reshape(A, B, C, S, S)[:, 1], x[:, 1].astype(np.float32))      # far outside: only the ego survives
    # non-finite poses: the kernel must neither fault nor invent occupancy
    bad = trans.copy()
    bad[0, 0, 1, 0, 3] = np.nan
    bad[0, 1, 0, 0, 0] = np.inf
    out = ops.warp_fuse(torch.from_numpy(ind).to(DEV), ops.warp_thetas(torch.from_numpy(bad)).to(DEV), torch.from_numpy(na).to(DEV), B, binary=True)
    got = out.cpu().numpy().reshape(A, B, C, S, S)
    assert set(np.unique(got)) <= {0.0, 1.0}
    assert (got[0, 0] >= x[0, 0]).all() and (got[0, 0] <= np.maximum(x[0, 0], x[2, 0])).all()   # agent 1's NaN pose contributes nothing to ego 0


@pytest.mark.parametrize("dt", ["bf16", "fp32"])
def test_colsum_groups_vs_torch(dt):
    """Grouped column sums (gradients of decoder_temp_embed / mask_token): ragged row counts, widths that do not fill a
    256-column block, time groups, both selector conventions; float64 torch sums of the same (rounded) values."""
    g = torch.Generator().manual_seed(5)
    for R, D, div, mod in ((40960, 256, 1024, 1), (3000, 132, 100, 3), (77, 384, 5, 2), (5000, 512, 1, 1)):
        src = torch.randn(R, D, generator=g)
        src = src.bfloat16() if dt == "bf16" else src
        sel = torch.randint(-1, 2, (R,), generator=g, dtype=torch.int32)
        grp = (torch.arange(R) // div) % mod
        for sel_t, nonneg in ((None, False), (sel, False), (sel, True)):
            keep = torch.ones(R, dtype=torch.bool) if sel_t is None else (sel >= 0 if nonneg else sel != 0)
            want = torch.zeros(mod, D, dtype=torch.float64)
            want.index_add_(0, grp[keep], src.double()[keep])
            got = ops.colsum_groups(src.to(DEV), div, mod, sel=None if sel_t is None else sel_t.to(DEV), sel_nonneg=nonneg)
            torch.testing.assert_close(got.cpu().double(), want, rtol=1e-5, atol=1e-3)
            again = ops.colsum_groups(src.to(DEV), div, mod, sel=None if sel_t is None else sel_t.to(DEV), sel_nonneg=nonneg)
            assert torch.equal(got, again)


@pytest.mark.parametrize("M,N", [(128, 1152), (128 * 33, 1536), (128 * 321, 1152), (40960, 2304)])
def test_gemm_w_in_registers_path(M, N):
    """Plain Linear with a bf16 output at K = 384, N % 192 == 0, M % 128 == 0, N >= 1152 takes the W-in-registers kernel
    (producer wave + W fragments resident in VGPRs): fewer row tiles than workgroups, uneven tiles per workgroup, the
    benched size; with and without bias; an identity A against an asymmetric W catches a wrong fragment or column map."""
    g = torch.Generator().manual_seed(M + N)
    K = 384
    a, w = _rand_lp((M, K), g, 0), _rand_lp((N, K), g, 0, K ** -0.5)
    bias = torch.randn(N, generator=g).to(DEV)
    ref = a.double() @ w.double().t()
    for b in (bias, None):
        lp, f32, _ = ops.gemm(a, w, bias=b, want_lp=True, want_f32=False)
        assert f32 is None
        torch.testing.assert_close(lp.double(), ref + (0 if b is None else b.double()), **_tol(0))
    eye = torch.zeros(M, K, device=DEV, dtype=torch.bfloat16)
    idx = torch.arange(M, device=DEV)
    eye[idx, idx % K] = 1
    lp, _, _ = ops.gemm(eye, w, want_lp=True, want_f32=False)
    assert torch.equal(lp, w.t()[idx % K])
    # the same kernel with the x-aux epilogue of fc2's input gradient
    from star_amd._lib import ACT_MULAUX
    aux = _rand_lp((M, N), g, 0)
    du, _, _ = ops.gemm(a, w, act=ACT_MULAUX, aux=aux)
    torch.testing.assert_close(du.double(), ref * aux.double(), **_tol(0))


def test_layernorm_deferred_folds_in_one_launch_are_bit_identical():
    """star_layernorm_bwd with dgamma = dbeta = NULL + star_layernorm_fold_batch over several calls (different M, D) against the
    immediate fold of the same calls."""
    g = torch.Generator().manual_seed(21)
    jobs, want = [], []
    for (m, d) in [(5000, 384), (333, 256), (40, 64), (9000, 384)]:
        x = torch.randn(m, d, generator=g).to(DEV)
        dy = torch.randn(m, d, generator=g).to(DEV).bfloat16()
        gamma = torch.randn(d, generator=g).to(DEV)
        mean, rstd = x.mean(1), (x.var(1, unbiased=False) + 1e-6).rsqrt()
        g1, l1, dg1, db1 = ops.layernorm_bwd(dy, x, gamma, mean, rstd)
        ws = torch.empty(L_mod.lib().star_layernorm_bwd_ws(d), dtype=torch.float32, device=DEV)
        dg2, db2 = torch.full((d,), float("nan"), device=DEV), torch.full((d,), float("nan"), device=DEV)
        g2, l2, _, _ = ops.layernorm_bwd(dy, x, gamma, mean, rstd, dgamma=dg2, dbeta=db2, defer_ws=ws)
        assert torch.equal(g1, g2) and torch.equal(l1, l2) and torch.isnan(dg2).all()      # not written before the fold
        jobs.append((ws, dg2, db2, m, d))
        want.append((dg1, db1))
    tab, n, max_d = ops.ln_fold_table(jobs, DEV)
    ops.ln_fold_batch(tab, n, max_d)
    for (ws, dg2, db2, m, d), (dg1, db1) in zip(jobs, want):
        assert torch.equal(dg1, dg2) and torch.equal(db1, db2)


@pytest.mark.parametrize("m,n,k", [(40960, 384, 384), (5120, 384, 1536), (1000, 384, 64), (777, 256, 256), (4096, 256, 1024), (1, 384, 384),
                                   (193, 384, 32)])
def test_gemm_res_ln_one_kernel_against_gemm_then_layernorm(m, n, k, monkeypatch):
    """star_gemm_res_ln against star_gemm (+ residual) followed by star_layernorm_fwd, and both against float64.  The float32
    residual-stream result must be bit-identical only up to the summation order of the MFMA chain (same products, the bias is
    added first instead of last): compared at float32 rounding; the bf16 LayerNorm output at one bf16 ulp of a unit-scale value."""
    g = torch.Generator().manual_seed(m + n + k)
    a = torch.randn(m, k, generator=g).to(DEV).bfloat16()
    w = (torch.randn(n, k, generator=g) * k ** -0.5).to(DEV).bfloat16()
    bias = torch.randn(n, generator=g).to(DEV)
    res = (torch.randn(m, n, generator=g) * 3 + 0.5).to(DEV)
    gamma, beta = torch.randn(n, generator=g).to(DEV), torch.randn(n, generator=g).to(DEV)
    assert ops.gemm_res_ln_supported(n, k, 0)
    _, x_ref, _ = ops.gemm(a, w, bias=bias, res=res, want_lp=False, want_f32=True)
    h_ref, mean_ref, rstd_ref = ops.layernorm_fwd(x_ref, gamma, beta, 1e-6, 0)
    for geo in (None, "0", "1", "2", "3", "4"):      # None: the library's own choice by shape
        if geo is None:
            monkeypatch.delenv("STAR_AMD_GEMM_LN_GEO", raising=False)
        else:
            monkeypatch.setenv("STAR_AMD_GEMM_LN_GEO", geo)
        x, h, mean, rstd = ops.gemm_res_ln(a, w, bias, res, gamma, beta, 1e-6)
        x64 = res.double() + a.double() @ w.double().t() + bias.double()
        assert (x - x64).abs().max().item() < 2e-5 * max(1.0, x64.abs().max().item())
        assert (x - x_ref).abs().max().item() < 1e-5 * max(1.0, x64.abs().max().item())
        mu64 = x64.mean(1, keepdim=True)
        rs64 = (x64.var(1, unbiased=False, keepdim=True) + 1e-6).rsqrt()
        h64 = (x64 - mu64) * rs64 * gamma.double() + beta.double()
        assert (mean - mu64[:, 0]).abs().max().item() < 1e-5
        assert ((rstd - rs64[:, 0]).abs() / rs64[:, 0]).max().item() < 1e-5
        tol = 2.0 ** -8 * (h64.abs() + 1.0)
        assert ((h.double() - h64).abs() <= tol).all()
        assert ((h.double() - h_ref.double()).abs() <= 2 * tol).all()      # each is within tol of the float64 value
    # in place on the residual stream (inference form), no statistics
    res2 = res.clone()
    x2, h2, mean2, rstd2 = ops.gemm_res_ln(a, w, bias, res2, gamma, beta, 1e-6, want_stats=False, out=res2)
    assert x2 is res2 and mean2 is None and torch.equal(x2, x) and torch.equal(h2, h)


def test_gemm_res_ln_rejects_unsupported_shapes():
    assert not ops.gemm_res_ln_supported(128, 384, 0) and not ops.gemm_res_ln_supported(384, 40, 0)
    # whole-row tiles: at small M the two launches are the faster path for the K = 1536 shape, and callers are told so
    assert ops.gemm_res_ln_prefers(40960, 384, 1536, 0) and not ops.gemm_res_ln_prefers(5120, 384, 1536, 0)
    assert ops.gemm_res_ln_prefers(5120, 384, 384, 0) and ops.gemm_res_ln_prefers(5120, 256, 1024, 0)
    assert not ops.gemm_res_ln_prefers(40960, 384, 1536, 1) and not ops.gemm_res_ln_prefers(0, 384, 384, 0)
    assert not ops.gemm_res_ln_supported(384, 384, 1)
    a = torch.zeros(8, 384, device=DEV, dtype=torch.bfloat16)
    w = torch.zeros(128, 384, device=DEV, dtype=torch.bfloat16)
    with pytest.raises(RuntimeError, match="star_gemm_res_ln"):
        ops.gemm_res_ln(a, w, torch.zeros(128, device=DEV), torch.zeros(8, 128, device=DEV), torch.ones(128, device=DEV),
                        torch.zeros(128, device=DEV), 1e-6)
