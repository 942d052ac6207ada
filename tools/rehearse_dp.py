#!/usr/bin/env python3
"""Rehearsal of the scene-parallel training step with 2 ranks SHARING one GPU over gloo (RCCL refuses two ranks on
one device).  usage (under torch.distributed.run): rehearse_dp.py nopar|sync|async
  nopar : two independent processes, no gradient exchange (does GPU sharing itself work?)
  sync  : blocking all_reduce of the whole flat gradient after backward
  async : SceneParallel with the buckets launched from the backward thread (the RCCL default)
  defer : SceneParallel with the buckets launched in finish() (the gloo default)
Known limit of the vehicle (not of the RCCL path, where every rank owns its GPU): with two processes sharing the GPU and gloo
staging device tensors through the host, a run occasionally stalls on its second step with one rank inside a device
synchronize (seen with and without the side stream, before and after unrelated changes); rerun.  The watchdog below ends it."""
import faulthandler
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist

from star_amd.models import multiagent_mae as mm
from star_amd.optim import FlatAdamW
from star_amd.parallel import SceneParallel
from star_amd.synthetic import make_batch
from star_amd.utils.CoModule import CoModule

faulthandler.dump_traceback_later(45, exit=True)
mode = sys.argv[1]
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo")
torch.manual_seed(0)
model = mm.amortized_ind_patch8_light(time_stamp=1, mask_method="complement").to("cuda").set_compute_dtype("bf16")
model.train()
opt = FlatAdamW(model, lr=1e-3, betas=(0.9, 0.95), weight_decay=0.05)
par = SceneParallel(model, opt, overlap=(mode == "async")) if mode in ("async", "defer") else None
if par is not None:
    par.broadcast_parameters(0)
cm = CoModule(model, opt, "ind_mae", parallel=par)
data = make_batch(2, 5, 1, seed=1 + rank, device="cuda")
for it in range(3):
    torch.manual_seed(123 + it * world + rank)
    if mode == "sync":
        loss, result, latent, ind = model(*cm._unpack(data), 2, mask_ratio=0.75)
        loss.backward()
        torch.cuda.synchronize()
        dist.all_reduce(opt.flat_g)
        opt.flat_g.div_(world)
        opt.step()
        lv = loss.item()
    else:
        lv, _, _ = cm.step_mae_completion(data, 2, 0.75, trainable=True)
    print("rank %d it %d loss %.6f" % (rank, it, lv), flush=True)
torch.cuda.synchronize()
# parameters must be identical on both ranks when gradients are exchanged
if mode != "nopar":
    mine = opt.flat_p.detach().clone()
    other = mine.clone()
    dist.broadcast(other, 0)
    print("rank %d params equal to rank 0: %s" % (rank, bool(torch.equal(mine, other))), flush=True)
dist.barrier()
dist.destroy_process_group()
