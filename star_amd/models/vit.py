"""Parameter containers with the timm-0.3.2 names the reference checkpoints use
(`patch_embed.proj`, `blocks.N.{norm1,attn.qkv,attn.proj,norm2,mlp.fc1,mlp.fc2}`).

These modules only OWN parameters (and create them in timm's order, so that a seeded construction
draws the same initial weights as the reference).  They compute nothing: the arithmetic of the whole
model runs in star_amd.engine through libstar_hip.so."""
import torch.nn as nn


def _no_forward(self, *a, **k):
    raise RuntimeError("star_amd modules are parameter containers; call the model (HIP engine) instead")


class Mlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.fc2 = nn.Linear(hidden, dim)

    forward = _no_forward


class Attention(nn.Module):
    def __init__(self, dim, num_heads, qkv_bias=True):
        super().__init__()
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)

    forward = _no_forward


class Block(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio=4.0, qkv_bias=True, qk_scale=None, norm_layer=nn.LayerNorm):
        super().__init__()
        assert qk_scale is None, "the HIP attention uses the default hd**-0.5 scale"
        self.norm1 = norm_layer(dim)
        self.attn = Attention(dim, num_heads, qkv_bias)
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))

    forward = _no_forward


class PatchEmbed(nn.Module):
    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768):
        super().__init__()
        self.img_size = (img_size, img_size)
        self.patch_size = (patch_size, patch_size)
        self.num_patches = (img_size // patch_size) ** 2
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)

    forward = _no_forward
