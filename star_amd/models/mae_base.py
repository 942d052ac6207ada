"""Base multi-agent masked-autoencoder: parameter inventory, initialisation and patch reshapes.

Mirrors the reference interface star/models/mae_base.py:257-406 (class name, constructor arguments,
parameter names and creation order -- so a seeded construction and a reference checkpoint both give the
same weights).  The compute of the path lives in star_amd.engine over libstar_hip.so."""
import torch
import torch.nn as nn

from ..utils.pos_embed import get_2d_sincos_pos_embed
from .vit import Block, PatchEmbed, _no_forward


class ConvPred(nn.Module):
    """Parameter container of the conv prediction head for patch 8 (reference mae_base.py:16-54): conv1/bn1 (C -> C/2),
    conv2/bn2 (-> C/4), conv3/bn3 (-> output_chans), every conv 3x3 pad 1, each behind a bilinear x2.  The arithmetic runs in
    star_amd.joint (channels-last, nine GEMM taps per convolution)."""

    def __init__(self, input_size=32, output_size=256, input_chans=512, output_chans=13):
        super().__init__()
        chans1 = input_chans // 2
        self.conv1 = nn.Conv2d(input_chans, chans1, kernel_size=3, stride=1, padding=1)
        self.bn1 = nn.BatchNorm2d(chans1)
        chans2 = chans1 // 2
        self.conv2 = nn.Conv2d(chans1, chans2, kernel_size=3, stride=1, padding=1)
        self.bn2 = nn.BatchNorm2d(chans2)
        self.conv3 = nn.Conv2d(chans2, output_chans, kernel_size=3, stride=1, padding=1)
        self.bn3 = nn.BatchNorm2d(output_chans)

    forward = _no_forward


class ConvPred16(nn.Module):
    """The same for patch 16 (reference mae_base.py:56-105): four stages, C -> C/2 -> C/4 -> C/8 -> output_chans."""

    def __init__(self, input_size=16, output_size=256, input_chans=512, output_chans=13):
        super().__init__()
        chans1 = input_chans // 2
        self.conv1 = nn.Conv2d(input_chans, chans1, kernel_size=3, stride=1, padding=1)
        self.bn1 = nn.BatchNorm2d(chans1)
        chans2 = chans1 // 2
        self.conv2 = nn.Conv2d(chans1, chans2, kernel_size=3, stride=1, padding=1)
        self.bn2 = nn.BatchNorm2d(chans2)
        chans3 = chans2 // 2
        self.conv3 = nn.Conv2d(chans2, chans3, kernel_size=3, stride=1, padding=1)
        self.bn3 = nn.BatchNorm2d(chans3)
        self.conv4 = nn.Conv2d(chans3, output_chans, kernel_size=3, stride=1, padding=1)
        self.bn4 = nn.BatchNorm2d(output_chans)

    forward = _no_forward


class MultiAgentMaskedAutoencoderViT(nn.Module):
    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=1024, depth=24, num_heads=16,
                 decoder_embed_dim=512, decoder_depth=8, decoder_num_heads=16, mlp_ratio=4., norm_layer=nn.LayerNorm,
                 decoder_head="mlp", norm_pix_loss=False, time_stamp=1, mask_method="random", inter_emb_dim=32):
        super().__init__()
        if decoder_head not in ("mlp", "conv3"):
            raise NotImplementedError("decoder head", decoder_head)
        # ---- encoder ----
        self.patch_embed = PatchEmbed(img_size, patch_size, in_chans, embed_dim)
        num_patches = self.patch_embed.num_patches
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, num_patches + 1, embed_dim), requires_grad=False)
        self.time_stamp = time_stamp
        self.temp_embed = nn.Parameter(torch.zeros(1, self.time_stamp, embed_dim))
        self.decoder_temp_embed = nn.Parameter(torch.zeros(1, self.time_stamp, 1, decoder_embed_dim))
        self.blocks = nn.ModuleList([Block(embed_dim, num_heads, mlp_ratio, qkv_bias=True, qk_scale=None, norm_layer=norm_layer)
                                     for _ in range(depth)])
        self.norm = norm_layer(embed_dim)
        self.compressor = nn.Sequential(nn.Linear(embed_dim, inter_emb_dim), nn.ReLU())
        # ---- decoder ----
        self.decompressor = nn.Linear(inter_emb_dim, embed_dim)
        self.decoder_embed = nn.Linear(embed_dim, decoder_embed_dim, bias=True)
        self.mask_token = nn.Parameter(torch.zeros(1, 1, decoder_embed_dim))
        self.decoder_pos_embed = nn.Parameter(torch.zeros(1, num_patches + 1, decoder_embed_dim), requires_grad=False)
        self.decoder_blocks = nn.ModuleList([Block(decoder_embed_dim, decoder_num_heads, mlp_ratio, qkv_bias=True, qk_scale=None,
                                                   norm_layer=norm_layer) for _ in range(decoder_depth)])
        self.decoder_norm = norm_layer(decoder_embed_dim)
        if decoder_head == "mlp":
            self.decoder_pred_occ = nn.Linear(decoder_embed_dim, patch_size ** 2 * in_chans, bias=True)
            self.decoder_pred_free = nn.Linear(decoder_embed_dim, patch_size ** 2 * in_chans, bias=True)
        elif patch_size == 8:       # mae_base.py:318-325: the head's 512 input channels are hard-coded in the reference
            self.decoder_pred_occ = ConvPred(input_size=32, output_size=256, input_chans=512, output_chans=13)
            self.decoder_pred_free = ConvPred(input_size=32, output_size=256, input_chans=512, output_chans=13)
        elif patch_size == 16:
            self.decoder_pred_occ = ConvPred16(input_size=16, output_size=256, input_chans=512, output_chans=13)
            self.decoder_pred_free = ConvPred16(input_size=16, output_size=256, input_chans=512, output_chans=13)
        else:
            raise NotImplementedError("not supported conv head for patch size", patch_size)

        self.norm_pix_loss = norm_pix_loss
        self.in_chans = in_chans
        self.mask_method = mask_method
        self.initialize_weights()
        self.num_agent = 0

    def initialize_weights(self):
        grid = int(self.patch_embed.num_patches ** .5)
        for table in (self.pos_embed, self.decoder_pos_embed):   # frozen sin-cos tables
            table.data.copy_(torch.from_numpy(get_2d_sincos_pos_embed(table.shape[-1], grid, cls_token=True)).float().unsqueeze(0))
        w = self.patch_embed.proj.weight.data
        torch.nn.init.xavier_uniform_(w.view([w.shape[0], -1]))   # patch embedding initialised like a Linear
        torch.nn.init.normal_(self.cls_token, std=.02)
        torch.nn.init.normal_(self.mask_token, std=.02)
        torch.nn.init.normal_(self.temp_embed, std=.02)           # decoder_temp_embed stays zero (Q5)
        self.apply(self._init_weights)

    def _init_weights(self, m):
        if isinstance(m, nn.Linear):
            torch.nn.init.xavier_uniform_(m.weight)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    def patchify(self, imgs):
        """[N,C,H,W] -> [N, L, p*p*C], vector order (py, px, c)."""
        p = self.patch_embed.patch_size[0]
        assert imgs.shape[2] == imgs.shape[3] and imgs.shape[2] % p == 0
        n, c, g = imgs.shape[0], imgs.shape[1], imgs.shape[2] // p
        return imgs.reshape(n, c, g, p, g, p).permute(0, 2, 4, 3, 5, 1).reshape(n, g * g, p * p * c)

    def unpatchify(self, x):
        """[N, L, p*p*C] -> [N,C,H,W]."""
        p = self.patch_embed.patch_size[0]
        g = int(x.shape[1] ** .5)
        assert g * g == x.shape[1]
        c = x.size(-1) // (p * p)
        assert p * p * c == x.size(-1)
        return x.reshape(x.shape[0], g, g, p, p, c).permute(0, 5, 1, 3, 2, 4).reshape(x.shape[0], c, g * p, g * p)
