"""Scene-parallel gradient averaging rehearsed on CPU: two processes, gloo backend, the same
SceneParallel bucketing / async all-reduce / finish() logic that runs over RCCL on the GPUs.
The flat buffers here are CPU tensors filled with rank-dependent values; the model itself is not run."""
import os
import types

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from star_amd import engine
from star_amd.models import multiagent_mae as mm
from star_amd.parallel import SceneParallel


def _fake_optimizer(model):
    cfg = model._cfg()
    named = dict(model.named_parameters())
    stages = engine.stages(cfg)
    ranges, off = [], 0
    for st in stages:
        n = sum((named[k].numel() + 3) // 4 * 4 for k in st)
        ranges.append((off, off + n))
        off += n
    offsets = {k: 0 for st in stages for k in st}      # names that live in the flat buffer (the values are not used here)
    return types.SimpleNamespace(is_star_flat=True, stage_ranges=ranges, flat_g=torch.zeros(off), flat_p=torch.zeros(off), total=off,
                                 offsets=offsets)


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    model = mm.AmortizedIndivMMAEViT(img_size=32, patch_size=8, in_chans=13, embed_dim=32, depth=2, num_heads=2, decoder_embed_dim=32,
                                     decoder_depth=2, decoder_num_heads=2, time_stamp=2, mask_method="complement")
    opt = _fake_optimizer(model)
    par = SceneParallel(model, opt, bucket_mb=0.05)
    assert model._on_grads_ready is not None and not par.native_avg
    # parameters: rank 0's copy wins
    opt.flat_p.fill_(float(rank + 1))
    with torch.no_grad():      # state outside the flat buffer (frozen tables, parameters no optimizer owns) travels too
        model.pos_embed.fill_(float(rank + 5))
        model.cls_token.fill_(float(rank + 7))
    par.broadcast_parameters(0)
    assert torch.all(opt.flat_p == 1.0) and torch.all(model.pos_embed == 5.0) and torch.all(model.cls_token == 7.0)
    # the stop/continue decision is taken together: one rank's non-finite loss stops every rank
    assert par.any_rank(rank == 1) is True and par.any_rank(False) is False
    # gradients: stage k of rank r holds r + k; the callbacks fire in backward order, as the engine does
    for k, (lo, hi) in enumerate(opt.stage_ranges):
        opt.flat_g[lo:hi] = rank + k
    for k in range(len(opt.stage_ranges)):
        model._on_grads_ready(k)
    par.finish()
    want = torch.zeros_like(opt.flat_g)
    for k, (lo, hi) in enumerate(opt.stage_ranges):
        want[lo:hi] = k + (world - 1) / 2.0
    ok = torch.equal(opt.flat_g, want) and len(par.pending) == 0
    # every element belongs to exactly one bucket
    cover = torch.zeros(opt.total)
    for lo, hi in par.close_at.values():
        cover[lo:hi] += 1
    ok = ok and bool(torch.all(cover == 1)) and len(par.close_at) > 1
    out[rank] = ok
    dist.destroy_process_group()


def test_scene_parallel_two_ranks_gloo():
    world = 2
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_worker, args=(world, 29533, out), nprocs=world, join=True)
        assert dict(out) == {0: True, 1: True}


def test_stage_list_covers_every_trainable_parameter_that_gets_a_gradient():
    for kw in (dict(time_stamp=1, mask_method="complement"), dict(time_stamp=3, mask_method="random"),
               dict(time_stamp=2, mask_method="complement", no_temp_emb=True, decode_singletemp=True)):
        m = mm.AmortizedIndivMMAEViT(img_size=32, patch_size=8, in_chans=13, embed_dim=32, depth=2, num_heads=2, decoder_embed_dim=32,
                                     decoder_depth=2, decoder_num_heads=2, **kw)
        names = [n for st in engine.stages(m._cfg()) for n in st]
        assert len(names) == len(set(names))
        trainable = {n for n, p in m.named_parameters() if p.requires_grad}
        assert set(names) <= trainable
        never = trainable - set(names)
        assert {"cls_token", "temp_embed"} <= never and never <= {"cls_token", "temp_embed", "mask_token", "decoder_temp_embed"}


def _ema_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from star_amd.models.VQSTAR import STARVectorQuantizerEMA
    torch.manual_seed(100 + rank)                      # every rank: its own scenes, its own randperm
    vq = STARVectorQuantizerEMA(16, 8, 0.25, 0.99)
    vq.initialize_from(torch.randn(4, 50, 8) + rank)
    mine = vq._ema_w.data.clone()
    ref = mine.clone()
    dist.broadcast(ref, 0)
    out[rank] = bool(torch.equal(mine, ref)) and vq.data_initialized == 1 and bool(mine.abs().sum() > 0)
    dist.destroy_process_group()


def test_ema_codebook_seeding_is_rank_zeros_on_every_rank():
    """every data-parallel replica must start the EMA codebook from ONE k-means seeding."""
    world = 2
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_ema_worker, args=(world, 29541, out), nprocs=world, join=True)
        assert dict(out) == {0: True, 1: True}
