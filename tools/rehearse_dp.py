#!/usr/bin/env python3
"""Rehearsal of the scene-parallel training step with 2 ranks SHARING one GPU over gloo (RCCL refuses two ranks on
one device).  usage (under torch.distributed.run): rehearse_dp.py nopar|sync|async|defer|equiv
  nopar : two independent processes, no gradient exchange (does GPU sharing itself work?)
  sync  : blocking all_reduce of the whole flat gradient after backward
  async : SceneParallel with the buckets handed over from the backward thread (the RCCL default)
  defer : SceneParallel with the buckets handed over in finish()
  equiv : the data-parallel contract itself -- rank r runs scenes [r*B/2, (r+1)*B/2) of ONE B-scene batch through
          SceneParallel (fp32 mode); the averaged gradient on every rank must equal the gradient of the whole batch on a
          single rank (computed by rank 0 afterwards, without the wrapper), and after the optimizer step the parameters too.
A watchdog dumps every thread's stack and exits if a mode takes longer than 90 s."""
import faulthandler
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist

from star_amd.models import multiagent_mae as mm
from star_amd.optim import FlatAdamW
from star_amd.parallel import SceneParallel
from star_amd.synthetic import make_batch, make_structured_batch
from star_amd.utils.CoModule import CoModule

faulthandler.dump_traceback_later(90, exit=True)
mode = sys.argv[1]
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo")


def scene_slice(data, B, A, lo, hi):
    """Scenes [lo, hi) of an agent-major batch (row a*B + b) as a batch of its own."""
    rows = torch.tensor([a * B + b for a in range(A) for b in range(lo, hi)])
    out = {}
    for k, v in data.items():
        if k in ("trans_matrices", "num_agent"):
            out[k] = v[lo:hi]
        else:
            out[k] = v[rows.to(v.device)] if v.dim() > 2 or v.shape[0] == A * B else v
    return out, rows


if mode == "equiv":
    B, A, T = 4, 3, 2
    torch.manual_seed(0)
    model = mm.amortized_ind_patch8_light(time_stamp=T, mask_method="complement")
    with torch.no_grad():
        model.decoder_temp_embed.normal_(std=0.02)
    model = model.to("cuda").set_compute_dtype("fp32")
    model.train()
    sd0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
    opt = FlatAdamW(model, lr=1e-3, betas=(0.9, 0.95), weight_decay=0.05)
    par = SceneParallel(model, opt, bucket_mb=8.0)
    par.broadcast_parameters(0)
    cm = CoModule(model, opt, "ind_mae", parallel=par)
    data = make_structured_batch(B, A, T, seed=3, device="cuda")
    noise = torch.rand(A * B, 1024, generator=torch.Generator().manual_seed(9)).cuda()
    half = B // world
    mine, rows = scene_slice(data, B, A, rank * half, (rank + 1) * half)
    model._noise_override = noise[rows.cuda()]
    loss, _, _ = cm.step_mae_completion(mine, half, 0.75, trainable=True)
    torch.cuda.synchronize()
    g_dp = opt.flat_g.detach().clone()
    p_dp = opt.flat_p.detach().clone()
    # the same on every rank?
    ref = g_dp.clone().cpu()
    dist.broadcast(ref, 0)
    same = bool(torch.equal(ref, g_dp.cpu()))
    losses = [None] * world
    dist.all_gather_object(losses, loss)
    ok = same
    if rank == 0:
        # single-rank reference on the whole batch, same weights, same noise
        model2 = mm.amortized_ind_patch8_light(time_stamp=T, mask_method="complement")
        model2.load_state_dict({k: v.cpu() for k, v in sd0.items()})
        model2 = model2.to("cuda").set_compute_dtype("fp32")
        model2.train()
        opt2 = FlatAdamW(model2, lr=1e-3, betas=(0.9, 0.95), weight_decay=0.05)
        cm2 = CoModule(model2, opt2, "ind_mae")
        model2._noise_override = noise
        loss2, _, _ = cm2.step_mae_completion(data, B, 0.75, trainable=True)
        torch.cuda.synchronize()
        eg = ((g_dp - opt2.flat_g).norm() / opt2.flat_g.norm()).item()
        ep = ((p_dp - opt2.flat_p).abs().max()).item()
        el = abs(sum(losses) / world - loss2)
        print("equiv: whole-gradient rel err %.3e, max param diff after the step %.3e, mean-of-rank-losses vs batch loss %.3e, "
              "ranks hold identical gradients: %s" % (eg, ep, el, same), flush=True)
        ok = same and eg < 2e-5 and el < 1e-6 and ep < 1e-4      # ep: AdamW divides by |g| + 1e-8, elements with |g| ~ eps amplify a 1e-10 difference
    flag = torch.tensor([1.0 if ok else 0.0])
    dist.broadcast(flag, 0)
    print("rank %d equiv %s" % (rank, "OK" if flag.item() > 0 and same else "FAILED"), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if flag.item() > 0 and same else 1)

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
        host = opt.flat_g.cpu()
        dist.all_reduce(host)
        opt.flat_g.copy_(host / world)
        opt.step()
        lv = loss.item()
    else:
        lv, _, _ = cm.step_mae_completion(data, 2, 0.75, trainable=True)
    print("rank %d it %d loss %.6f" % (rank, it, lv), flush=True)
torch.cuda.synchronize()
# parameters must be identical on both ranks when gradients are exchanged
if mode != "nopar":
    mine = opt.flat_p.detach().cpu()
    other = mine.clone()
    dist.broadcast(other, 0)
    print("rank %d params equal to rank 0: %s" % (rank, bool(torch.equal(mine, other))), flush=True)
dist.barrier()
dist.destroy_process_group()
