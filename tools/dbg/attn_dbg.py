import sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from star_amd import ops
torch.manual_seed(0)
def run(NB, L, H, hd):
    g = torch.Generator().manual_seed(7 * NB + L + H + hd)
    qkv = (torch.randn(NB * L, 3 * H * hd, generator=g) * 1.5).cuda().bfloat16()
    dout = torch.randn(NB * L, H * hd, generator=g).cuda().bfloat16()
    out, lse = ops.attn_fwd(qkv, NB, L, H, hd)
    two = ops.attn_bwd(qkv, out, dout, lse, NB, L, H, hd, fused=False).float().reshape(NB, L, 3, H, hd)
    one = ops.attn_bwd(qkv, out, dout, lse, NB, L, H, hd, fused=True).float().reshape(NB, L, 3, H, hd)
    d = (one - two).abs()
    print("shape", NB, L, H, hd, "max |dq| err %.3g  dk %.3g  dv %.3g   (|dq| max %.3g)" % (d[:, :, 0].max(), d[:, :, 1].max(), d[:, :, 2].max(), two[:, :, 0].abs().max()))
    bad = d[:, :, 0] > 0.05
    if bad.any():
        print("  bad per nb", bad.sum((1, 2, 3)).tolist(), "per head", bad.sum((0, 1, 3)).tolist())
        print("  bad per q-block of 32", bad.reshape(NB, -1, 32 if L >= 32 else L, H, hd).sum((0, 2, 3, 4)).tolist() if L % 32 == 0 else "")
        print("  bad per d", bad.sum((0, 1, 2)).tolist())
        print("  bad per q%32", bad.reshape(NB, -1, 32, H, hd).sum((0, 1, 3, 4)).tolist() if L % 32 == 0 else "")
for s in [(1, 64, 1, 32), (1, 64, 1, 16), (3, 64, 3, 32), (1, 128, 2, 32), (1, 1024, 1, 32), (1, 1024, 2, 16)]:
    run(*s)
    torch.cuda.synchronize()
