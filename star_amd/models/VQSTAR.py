"""VQ-STAR: the STAR completion model with a vector quantiser between compressor and decompressor.

Mirrors the reference interface star/models/VQSTAR.py:14-480 (`VQSTARViT`, constructor arguments, `forward` /
`inference` signatures and return tuples, factory `vq_amo_individual_star_patch8_dec256d4b` and its alias `vqstar`
that completion/train_completion.py:93-98 calls) and the state-dict key `_vq_star._embedding.weight` of
STARVectorQuantizer (star/models/mae_base.py:107-120), or with `decay` > 0 of STARVectorQuantizerEMA (mae_base.py:176-253: `_embedding.weight`,
`_ema_w`, `_ema_cluster_size`).  Everything else of the model is the amortised-individual
path (the reference's VQSTAR.py repeats those methods verbatim); the quantiser runs in libstar_hip.so
(star_vq_assign / star_vq_finish / star_vq_bwd; star_vq_code_sums / star_vq_ema_update for the EMA codebook)."""
from functools import partial

import torch
import torch.nn as nn

from .. import ops
from .multiagent_mae import AmortizedIndivMMAEViT


class STARVectorQuantizer(nn.Module):
    """Parameter holder with the reference's names: codebook `_embedding.weight` [K, E] ~ N(0, 1) (nn.Embedding's
    default; the reference leaves it untouched -- it is created after initialize_weights ran)."""

    def __init__(self, num_embeddings, embedding_dim, commitment_cost):
        super().__init__()
        self._embedding_dim = embedding_dim
        self._num_embeddings = num_embeddings
        self._embedding = nn.Embedding(num_embeddings, embedding_dim)
        self._commitment_cost = commitment_cost
        self.data_initialized = 1            # mae_base.py:120: the k-means re-initialisation never triggers


class STARVectorQuantizerEMA(nn.Module):
    """State holder of the EMA quantiser (mae_base.py:176-195): codebook ~ N(0, 1), `_ema_cluster_size` buffer, `_ema_w`.
    The reference re-creates `_embedding.weight` and `_ema_w` as fresh nn.Parameters in every training forward, so no
    optimizer ever updates them; here they are persistent tensors with requires_grad=False that star_vq_ema_update
    rewrites in place (same state-dict keys and values).  `_ema_w` is uninitialised memory in the reference until the
    one-off k-means initialisation of the first training forward (mae_base.py:200-208) fills it; zeros here."""

    def __init__(self, num_embeddings, embedding_dim, commitment_cost, decay, epsilon=1e-5):
        super().__init__()
        self._embedding_dim = embedding_dim
        self._num_embeddings = num_embeddings
        self._embedding = nn.Embedding(num_embeddings, embedding_dim)
        self._embedding.weight.data.normal_()
        self._embedding.weight.requires_grad_(False)
        self._commitment_cost = commitment_cost
        self.register_buffer("_ema_cluster_size", torch.zeros(num_embeddings))
        self._ema_w = nn.Parameter(torch.zeros(num_embeddings, embedding_dim), requires_grad=False)
        self._decay = decay
        self._epsilon = epsilon
        self.data_initialized = 0

    def initialize_from(self, latent):
        """mae_base.py:200-208: k-means (scipy kmeans2, minit="points") over the shuffled latent rows of the first
        training batch seeds `_ema_w`; runs once, on the host, like the reference's."""
        import torch.distributed as dist
        multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        if not multi or dist.get_rank() == 0:
            from scipy.cluster.vq import kmeans2
            flat = latent.detach().reshape(-1, self._embedding_dim)
            rp = torch.randperm(flat.size(0), device=flat.device)
            centers, _ = kmeans2(flat[rp].cpu().numpy(), self._num_embeddings, minit="points")
            self._ema_w.data.copy_(torch.from_numpy(centers).to(self._ema_w.dtype))
        if multi:
            # data-parallel replicas must start from ONE seeding (rank 0's scenes): every rank has its own latents, its own
            # randperm and its own k-means run, and nothing else ever synchronises `_ema_w` (it is no optimizer parameter)
            if dist.get_backend() == "nccl" or not self._ema_w.is_cuda:
                dist.broadcast(self._ema_w.data, 0)
            else:
                host = self._ema_w.data.cpu()
                dist.broadcast(host, 0)
                self._ema_w.data.copy_(host)
        self.data_initialized = 1


class VQSTARViT(AmortizedIndivMMAEViT):
    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=1024, depth=24, num_heads=16, decoder_embed_dim=512,
                 decoder_depth=8, decoder_num_heads=16, mlp_ratio=4., norm_layer=nn.LayerNorm, decoder_head="mlp", norm_pix_loss=False,
                 time_stamp=1, mask_method="complement", encode_partial=False, no_temp_emb=False, decode_singletemp=False, decay=0.,
                 commitment_cost=0.25, num_vq_embeddings=512, vq_embedding_dim=64):
        super().__init__(img_size=img_size, patch_size=patch_size, in_chans=in_chans, embed_dim=embed_dim, depth=depth,
                         num_heads=num_heads, decoder_embed_dim=decoder_embed_dim, decoder_depth=decoder_depth,
                         decoder_num_heads=decoder_num_heads, mlp_ratio=mlp_ratio, norm_layer=norm_layer, decoder_head=decoder_head,
                         norm_pix_loss=norm_pix_loss, time_stamp=time_stamp, mask_method=mask_method, encode_partial=encode_partial,
                         no_temp_emb=no_temp_emb, decode_singletemp=decode_singletemp, inter_emb_dim=vq_embedding_dim)
        if vq_embedding_dim not in (8, 16, 32, 64, 128):
            raise NotImplementedError("vq_embedding_dim must be one of 8, 16, 32, 64, 128 (got %d)" % vq_embedding_dim)
        if decay > 0.0:      # VQSTAR.py:85-90
            self._vq_star = STARVectorQuantizerEMA(num_vq_embeddings, vq_embedding_dim, commitment_cost, decay)
        else:
            self._vq_star = STARVectorQuantizer(num_vq_embeddings, vq_embedding_dim, commitment_cost)

    def _frozen_tables(self):
        tables = super()._frozen_tables()
        vq = self._vq_star
        if isinstance(vq, STARVectorQuantizerEMA):
            tables.update({"_vq_star._embedding.weight": vq._embedding.weight, "_vq_star._ema_w": vq._ema_w,
                           "_vq_star._ema_cluster_size": vq._ema_cluster_size})
        return tables

    def _inputs(self, *args, **kwargs):
        inputs = super()._inputs(*args, **kwargs)
        vq = self._vq_star
        if inputs["vq_update"] and isinstance(vq, STARVectorQuantizerEMA) and not vq.data_initialized:
            # the k-means seeding needs this batch's latent: one extra forward without the update, once per training run
            inputs["vq_update"] = False
            with torch.no_grad():
                latent = self._run(inputs)[2]
            vq.initialize_from(latent)
            inputs["vq_update"] = True
        return inputs

    def forward(self, imgs1, imgs_next, teacher, trans_matrices, num_agent_tensor, batch_size, mask_ratio=0.75):
        """-> (loss = vq + reconstruction, result, mask, ind_result, perplexity)   (VQSTAR.py:421-444)"""
        loss, result, _latent, ind, mask, perplexity, _ = self._run(
            self._inputs(imgs1, imgs_next, trans_matrices, num_agent_tensor, batch_size, mask_ratio, infer=False))
        return loss, result, mask, ind, perplexity.reshape(())

    def inference(self, imgs1, imgs_next, teacher, trans_matrices, num_agent_tensor, batch_size, mask_ratio=0.75):
        """-> (loss, result, ind_result, perplexity, encodings [rows, K] one-hot)   (VQSTAR.py:446-470)"""
        loss, result, _latent, ind, _mask, perplexity, vq_idx = self._run(
            self._inputs(imgs1, imgs_next, trans_matrices, num_agent_tensor, batch_size, mask_ratio, infer=True))
        return loss, result, ind, perplexity.reshape(()), ops.onehot_rows(vq_idx, self._vq_star._num_embeddings)


def vq_amo_individual_star_patch8_dec256d4b(**kwargs):
    return VQSTARViT(img_size=kwargs.pop("img_size", 256), patch_size=8, in_chans=13, embed_dim=384, depth=6, num_heads=12,
                     decoder_embed_dim=256, decoder_depth=4, decoder_num_heads=16, mlp_ratio=4, norm_layer=partial(nn.LayerNorm, eps=1e-6),
                     decoder_head="mlp", **kwargs)


vqstar = vq_amo_individual_star_patch8_dec256d4b
