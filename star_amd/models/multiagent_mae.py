"""STAR temporal-amortised individual-reconstruction model on the MI355X HIP engine.

Mirrors the reference interface star/models/multiagent_mae.py:684-1287 for the hot path: class
`AmortizedIndivMMAEViT` (same constructor arguments, `forward` / `inference` signatures and return tuples,
attributes `patch_h/patch_w/time_stamp`), and the factory functions + aliases that
`multiagent_mae.__dict__[args.mae_model]` selects (train_completion.py:90).  The joint-fusion and non-amortised
model (`AmortizedFusionMMAEViT`, the reference's amortized_joint_* factories) runs through star_amd.joint; the older
non-amortised constructors are outside the path and raise NotImplementedError.
"""
import os
from functools import partial

import torch
import torch.nn as nn

from .. import engine
from .._lib import BF16, BF16X3, F32
from .mae_base import MultiAgentMaskedAutoencoderViT


_EXTRAS = ("mask", "perplexity", "vq_indices")     # after (loss, result, latent, ind): what the VQ-STAR variant also returns


class _StarFunction(torch.autograd.Function):
    """One autograd node for the whole model: forward runs the HIP forward chain and keeps the
    activations, backward runs the HIP backward chain and hands one gradient per parameter."""

    @staticmethod
    def forward(ctx, model, inputs, names, *params):
        P = dict(zip(names, params))
        P.update(model._frozen_tables())
        out, saved = engine.forward(model._cfg(), P, model._weights, model._dt(), train=True, **inputs)
        ctx.model, ctx.names, ctx.saved, ctx.P = model, names, saved, P
        ctx.set_materialize_grads(False)      # no zero-filled gradients for result / latent / ind (400 MB of fills per step)
        extras = tuple(out[k] for k in _EXTRAS if k in out)
        ctx.mark_non_differentiable(out["result"], out["latent"], out["ind"], *extras)
        return (out["loss"].reshape(()), out["result"], out["latent"], out["ind"]) + extras

    @staticmethod
    def backward(ctx, gloss, *_unused):
        model = ctx.model
        if gloss is None:      # the loss took no gradient (materialisation is off): nothing to propagate
            return (None, None, None) + (None,) * len(ctx.names)
        grads = engine.backward(model._cfg(), ctx.P, model._weights, model._dt(), ctx.saved, G=model._grad_views,
                                on_ready=model._on_grads_ready)
        ctx.saved = None
        if model._grad_views is not None:
            # flat-gradient mode (star_amd.optim.FlatAdamW): gradients were written straight into the flat
            # buffer that param.grad already views; the upstream scale is applied inside the optimizer kernel
            model._grad_scale = gloss.detach().reshape(1).float()
            return (None, None, None) + (None,) * len(ctx.names)
        outs = [grads.get(n) for n in ctx.names]
        live = [g for g in outs if g is not None]
        torch._foreach_mul_(live, gloss)          # upstream scale (loss scaler); exact for powers of two
        return (None, None, None) + tuple(outs)


class AmortizedIndivMMAEViT(MultiAgentMaskedAutoencoderViT):
    """Temporal amortised version: encode every timestamp alone, sub-sample tokens, compress to 32-d,
    mix the timestamps in the decoder, predict 2-class occupancy, fuse neighbours into the ego frame."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=1024, depth=24, num_heads=16,
                 decoder_embed_dim=512, decoder_depth=8, decoder_num_heads=16, mlp_ratio=4., norm_layer=nn.LayerNorm,
                 decoder_head="mlp", norm_pix_loss=False, time_stamp=1, mask_method="random", encode_partial=False,
                 no_temp_emb=False, decode_singletemp=False, inter_emb_dim=32):
        super().__init__(img_size=img_size, patch_size=patch_size, in_chans=in_chans, embed_dim=embed_dim, depth=depth,
                         num_heads=num_heads, decoder_embed_dim=decoder_embed_dim, decoder_depth=decoder_depth,
                         decoder_num_heads=decoder_num_heads, mlp_ratio=mlp_ratio, norm_layer=norm_layer,
                         decoder_head=decoder_head, norm_pix_loss=norm_pix_loss, time_stamp=time_stamp, mask_method=mask_method,
                         inter_emb_dim=inter_emb_dim)
        self.patch_h = 0
        self.patch_w = 0
        if mask_method not in ("random", "complement"):
            raise NotImplementedError(mask_method)
        self.encode_partial = encode_partial
        self.decode_singletemp = decode_singletemp
        self.no_temp_emb = no_temp_emb
        # ---- engine state (not parameters) ----
        self._weights = engine.Weights()
        self._config = None
        self._compute_dtype = os.environ.get("STAR_AMD_DTYPE", "bf16")
        self._grad_views = None          # name -> view into a flat gradient buffer (star_amd.optim.FlatAdamW)
        self._grad_scale = None          # device scalar: upstream gradient of the loss in flat-gradient mode
        self._on_grads_ready = None      # callback(stage) from the backward chain (bucketed all-reduce)
        self._noise_override = None      # tests: feed the mask noise instead of drawing it
        self._taps = None                # tests: dict that receives intermediate tensors
        self._theta_override = None      # star_amd.graph: static device buffer with the warp table (replaces trans_matrices)

    # ---- engine plumbing --------------------------------------------------------------------------------
    def set_compute_dtype(self, name):
        """'bf16' (bf16 storage + MFMA, fp32 accumulation; default), 'fp32' (the reference's numerics) or 'bf16x3' (inference only:
        bf16 hi/lo pairs, three bf16 MFMA products per multiply -- float32-grade decisions at bf16-pipe speed)."""
        assert name in ("bf16", "fp32", "bf16x3"), name
        self._compute_dtype = name
        return self

    def _dt(self):
        return {"bf16": BF16, "fp32": F32, "bf16x3": BF16X3}[self._compute_dtype]

    def _cfg(self):
        if self._config is None:
            self._config = engine.Config(self)
        return self._config

    def _frozen_tables(self):
        return {"pos_embed": self.pos_embed, "decoder_pos_embed": self.decoder_pos_embed}

    def _inputs(self, imgs1, imgs_next, trans_matrices, num_agent_tensor, batch_size, mask_ratio, infer):
        if not imgs1.is_cuda:
            raise RuntimeError("star_amd runs on an MI355X only: move the model and its inputs to 'cuda' "
                               "(there is no CPU path; the CPU oracle lives under oracle/ for tests)")
        assert imgs1.shape[0] % batch_size == 0, (imgs1.size(), batch_size)        # multiagent_mae.py:878
        p = self.patch_embed.patch_size[0]
        self.patch_h = self.patch_w = imgs1.shape[2] // p                          # multiagent_mae.py:1090
        self.num_agent = imgs1.shape[0] // batch_size
        return dict(imgs1=imgs1.float(), imgs_next=imgs_next.float() if imgs_next.dim() > 2 else imgs_next,
                    trans_matrices=trans_matrices, num_agent=num_agent_tensor, batch=batch_size, mask_ratio=mask_ratio,
                    infer=infer, noise=self._noise_override, taps=self._taps, vq_update=self.training, theta=self._theta_override)

    def _named_trainable(self):
        names, params = [], []
        for n, prm in self.named_parameters():
            if prm.requires_grad:
                names.append(n)
                params.append(prm)
        return tuple(names), params

    def _run(self, inputs):
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            names, params = self._named_trainable()
            return _StarFunction.apply(self, inputs, names, *params)
        P = {n: prm for n, prm in self.named_parameters()}
        P.update(self._frozen_tables())
        out, _ = engine.forward(self._cfg(), P, self._weights, self._dt(), train=False, **inputs)
        return (out["loss"].reshape(()), out["result"], out["latent"], out["ind"]) + tuple(out[k] for k in _EXTRAS if k in out)

    # ---- reference interface ----------------------------------------------------------------------------
    def forward(self, imgs1, imgs_next, teacher, trans_matrices, num_agent_tensor, batch_size, mask_ratio=0.75):
        """-> (loss, result [A*B,C,H,W] float32 in {0,1}, latent [A*B*T, keep, 32], ind_result int64).
        The loss target is the agent's own input, `teacher` is unused (reference Q4); the ego slot of the fusion
        uses the agent's own prediction (late_fusion)."""
        return self._run(self._inputs(imgs1, imgs_next, trans_matrices, num_agent_tensor, batch_size, mask_ratio, infer=False))[:4]

    def inference(self, imgs1, imgs_next, teacher, trans_matrices, num_agent_tensor, batch_size, mask_ratio=0.75):
        """Same tuple; the ego slot of the fusion uses the agent's own INPUT (ego_late_fusion)."""
        return self._run(self._inputs(imgs1, imgs_next, trans_matrices, num_agent_tensor, batch_size, mask_ratio, infer=True))[:4]


class _JointFunction(torch.autograd.Function):
    """One autograd node for the joint model (star_amd.joint): gradients come back per parameter name."""

    @staticmethod
    def forward(ctx, model, inputs, names, *params):
        from .. import joint
        P = dict(zip(names, params))
        P.update(model._frozen_tables())
        P.update({n: b for n, b in model.named_buffers()})
        out, saved = joint.forward(model._cfg(), P, model._weights, model._dt(), train=True, **inputs)
        ctx.model, ctx.names, ctx.saved, ctx.P = model, names, saved, P
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(out["result"], out["mask"], out["pred"])
        return out["loss"].reshape(()), out["result"], out["mask"], out["pred"]

    @staticmethod
    def backward(ctx, gloss, *_unused):
        from .. import joint
        if gloss is None:
            return (None, None, None) + (None,) * len(ctx.names)
        grads = joint.backward(ctx.model._cfg(), ctx.P, ctx.model._weights, ctx.model._dt(), ctx.saved)
        ctx.saved = None
        outs = [grads.get(n) for n in ctx.names]
        outs = [None if g is None else g.reshape(p.shape) for g, p in zip(outs, [ctx.P[n] for n in ctx.names])]
        torch._foreach_mul_([g for g in outs if g is not None], gloss)
        return (None, None, None) + tuple(outs)


class AmortizedFusionMMAEViT(MultiAgentMaskedAutoencoderViT):
    """Temporal amortised JOINT model (reference multiagent_mae.py:347-681): encode every timestamp alone, transmit a token
    subset, warp the neighbours' decoder-side feature maps into the ego frame and fuse them by sum + threshold BEFORE the
    decoder, predict with two conv heads, cross entropy against the TEACHER map.  forward -> (loss, result int64 [A*B,C,H,W],
    mask, pred [A*B,2,C,H,W]); the reference defines no `inference` for it."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=1024, depth=24, num_heads=16, decoder_embed_dim=512,
                 decoder_depth=8, decoder_num_heads=16, mlp_ratio=4., norm_layer=nn.LayerNorm, decoder_head="conv3", norm_pix_loss=False,
                 time_stamp=1, mask_method="random"):
        super().__init__(img_size=img_size, patch_size=patch_size, in_chans=in_chans, embed_dim=embed_dim, depth=depth, num_heads=num_heads,
                         decoder_embed_dim=decoder_embed_dim, decoder_depth=decoder_depth, decoder_num_heads=decoder_num_heads,
                         mlp_ratio=mlp_ratio, norm_layer=norm_layer, decoder_head=decoder_head, norm_pix_loss=norm_pix_loss,
                         time_stamp=time_stamp, mask_method=mask_method)
        self.patch_h = 0
        self.patch_w = 0
        if mask_method not in ("random", "complement"):
            raise NotImplementedError(mask_method)
        self.encode_partial = self.no_temp_emb = self.decode_singletemp = False      # flags of the individual model only
        self._weights = engine.Weights()
        self._config = None
        self._compute_dtype = os.environ.get("STAR_AMD_DTYPE", "bf16")
        self._noise_override = None
        self._taps = None

    set_compute_dtype = AmortizedIndivMMAEViT.set_compute_dtype
    _dt = AmortizedIndivMMAEViT._dt
    _cfg = AmortizedIndivMMAEViT._cfg
    _frozen_tables = AmortizedIndivMMAEViT._frozen_tables
    _named_trainable = AmortizedIndivMMAEViT._named_trainable

    def forward(self, imgs1, imgs_next, teacher, trans_matrices, num_agent_tensor, batch_size, mask_ratio=0.75):
        if not imgs1.is_cuda:
            raise RuntimeError("star_amd runs on an MI355X only: move the model and its inputs to 'cuda' (there is no CPU path)")
        assert imgs1.shape[0] % batch_size == 0, (imgs1.size(), batch_size)             # multiagent_mae.py:513
        from .. import joint
        p = self.patch_embed.patch_size[0]
        self.patch_h = self.patch_w = imgs1.shape[2] // p
        self.num_agent = imgs1.shape[0] // batch_size
        inputs = dict(imgs1=imgs1.float(), imgs_next=imgs_next.float() if imgs_next.dim() > 2 else imgs_next, teacher=teacher.float(),
                      trans_matrices=trans_matrices, num_agent=num_agent_tensor, batch=batch_size, mask_ratio=mask_ratio,
                      training=self.training, noise=self._noise_override, taps=self._taps)
        if torch.is_grad_enabled() and any(p_.requires_grad for p_ in self.parameters()):
            names, params = self._named_trainable()
            return _JointFunction.apply(self, inputs, names, *params)
        P = {n: prm for n, prm in self.named_parameters()}
        P.update(self._frozen_tables())
        P.update({n: b for n, b in self.named_buffers()})
        out, _ = joint.forward(self._cfg(), P, self._weights, self._dt(), train=False, **inputs)
        return out["loss"].reshape(()), out["result"], out["mask"], out["pred"]


def _amo_fusion(patch, depth, ddepth, **kwargs):      # multiagent_mae.py:1186-1213
    return AmortizedFusionMMAEViT(img_size=kwargs.pop("img_size", 256), patch_size=patch, in_chans=13, embed_dim=768, depth=depth, num_heads=12,
                                  decoder_embed_dim=512, decoder_depth=ddepth, decoder_num_heads=16, mlp_ratio=4,
                                  norm_layer=partial(nn.LayerNorm, eps=1e-6), decoder_head="conv3", **kwargs)


def amo_fusion_bev_multi_mae_vit_base_patch8_dec512d6b(**kwargs):
    return _amo_fusion(8, 6, 6, **kwargs)


def amo_fusion_bev_multi_mae_vit_base_patch16_dec512d6b(**kwargs):
    return _amo_fusion(16, 6, 6, **kwargs)


def amo_fusion_bev_multi_mae_vit_base_patch8_dec512d8b(**kwargs):
    return _amo_fusion(8, 12, 8, **kwargs)


def amo_fusion_bev_multi_mae_vit_base_patch16_dec512d8b(**kwargs):
    return _amo_fusion(16, 12, 8, **kwargs)


# ---- factory functions (multiagent_mae.py:1129-1184) -------------------------------------------------------
def _amo(patch, dim, depth, ddim, ddepth, **kwargs):
    return AmortizedIndivMMAEViT(img_size=kwargs.pop("img_size", 256), patch_size=patch, in_chans=13, embed_dim=dim, depth=depth,
                                 num_heads=12, decoder_embed_dim=ddim, decoder_depth=ddepth, decoder_num_heads=16, mlp_ratio=4,
                                 norm_layer=partial(nn.LayerNorm, eps=1e-6), decoder_head="mlp", **kwargs)


def amo_individual_bev_multi_mae_vit_base_patch8_dec512d4b(**kwargs):
    return _amo(8, 768, 6, 512, 4, **kwargs)


def amo_individual_bev_multi_mae_vit_base_patch8_dec256d4b(**kwargs):
    return _amo(8, 384, 6, 256, 4, **kwargs)


def amo_individual_bev_multi_mae_vit_base_patch16_dec256d4b(**kwargs):
    return _amo(16, 384, 6, 256, 4, **kwargs)


def amo_individual_bev_multi_mae_vit_base_patch32_dec256d4b(**kwargs):
    return _amo(32, 384, 6, 256, 4, **kwargs)


def amo_individual_bev_multi_mae_vit_base_patch8_dec512d8b(**kwargs):
    return _amo(8, 768, 12, 512, 8, **kwargs)


def amo_individual_bev_multi_mae_vit_base_patch16_dec512d8b(**kwargs):
    return _amo(16, 768, 12, 512, 8, **kwargs)


def amo_individual_bev_multi_mae_vit_base_patch32_dec512d8b(**kwargs):
    return _amo(32, 768, 12, 512, 8, **kwargs)


def amo_individual_bev_multi_mae_vit_base_patch4_dec512d8b(**kwargs):
    return _amo(4, 768, 12, 512, 8, **kwargs)


def _outside_hot_path(name):
    def ctor(**kwargs):
        raise NotImplementedError(name + ": the non-amortised STAR variants are outside the accelerated path (SURVEY.md section 2 "
                                  "#11-12: older models, one of them broken in the reference); use an amortized_ind_* or amortized_joint_* model")
    ctor.__name__ = name
    return ctor


for _n in ("individual_bev_multi_mae_vit_base_patch8_dec512d8b", "fusion_bev_multi_mae_vit_base_patch16_dec512d8b",
           "fusion_bev_multi_mae_vit_base_patch8_dec512d8b", "fusion_bev_multi_mae_vit_base_patch1_dec512d8b",
           "fusion_bev_multi_mae_vit_base_patch32_dec512d8b", "bev_multi_mae_vit_base_patch16_dec512d8b",
           "fusion_bev_multi_mae_vit_base_patch16_dec1024d8b"):
    globals()[_n] = _outside_hot_path(_n)

# aliases (multiagent_mae.py:1266-1287)
bev_mae_vit_base_patch16 = bev_multi_mae_vit_base_patch16_dec512d8b  # noqa: F821
joint_bev_mae_vit_base_patch16 = fusion_bev_multi_mae_vit_base_patch16_dec512d8b  # noqa: F821
joint_bev_mae_vit_base_patch8 = fusion_bev_multi_mae_vit_base_patch8_dec512d8b  # noqa: F821
joint_bev_mae_vit_base_patch1 = fusion_bev_multi_mae_vit_base_patch1_dec512d8b  # noqa: F821
joint_bev_mae_vit_base_patch32 = fusion_bev_multi_mae_vit_base_patch32_dec512d8b  # noqa: F821
joint_bev_mae_vit_base_patch16_dec1024 = fusion_bev_multi_mae_vit_base_patch16_dec1024d8b  # noqa: F821
ind_bev_mae_vit_base_patch8 = individual_bev_multi_mae_vit_base_patch8_dec512d8b  # noqa: F821
amortized_ind_patch8 = amo_individual_bev_multi_mae_vit_base_patch8_dec512d8b
amortized_ind_patch8_shallow = amo_individual_bev_multi_mae_vit_base_patch8_dec512d4b
amortized_ind_patch8_light = amo_individual_bev_multi_mae_vit_base_patch8_dec256d4b
amortized_ind_patch16 = amo_individual_bev_multi_mae_vit_base_patch16_dec512d8b
amortized_ind_patch32 = amo_individual_bev_multi_mae_vit_base_patch32_dec512d8b
amortized_ind_patch4 = amo_individual_bev_multi_mae_vit_base_patch4_dec512d8b
amortized_joint_patch8_shallow = amo_fusion_bev_multi_mae_vit_base_patch8_dec512d6b
amortized_joint_patch16_shallow = amo_fusion_bev_multi_mae_vit_base_patch16_dec512d6b
amortized_joint_patch8 = amo_fusion_bev_multi_mae_vit_base_patch8_dec512d8b
amortized_joint_patch16 = amo_fusion_bev_multi_mae_vit_base_patch8_dec512d8b
amortized_ind_patch16_light = amo_individual_bev_multi_mae_vit_base_patch16_dec256d4b
amortized_ind_patch32_light = amo_individual_bev_multi_mae_vit_base_patch32_dec256d4b
