#!/usr/bin/env python3
"""Where the HOST spends a training step: time to enqueue the forward, the backward, to wait for the loss, to enqueue the
optimizer -- against the GPU time per step.  (GPU box only.)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from star_amd.models import multiagent_mae as mm
from star_amd.optim import FlatAdamW
from star_amd.synthetic import make_batch
from star_amd.utils.CoModule import CoModule

dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = mm.amortized_ind_patch8_light(time_stamp=1, mask_method="complement").to(dev).set_compute_dtype("bf16")
model.train()
opt = FlatAdamW(model, lr=1e-3, betas=(0.9, 0.95), weight_decay=0.05)
cm = CoModule(model, opt, "ind_mae")
data = make_batch(8, 5, 1, seed=1, device=dev)
for _ in range(5):
    cm.step_mae_completion(data, 8, 0.75, trainable=True)
torch.cuda.synchronize()
N = 20
acc = dict(fwd=0.0, bwd=0.0, wait=0.0, opt=0.0)
pinned = torch.empty((), dtype=torch.float32).pin_memory()
ev = torch.cuda.Event()
t_all = time.perf_counter()
for _ in range(N):
    t0 = time.perf_counter()
    loss, result, _, ind = model(*cm._unpack(data), 8, mask_ratio=0.75)
    pinned.copy_(loss.detach(), non_blocking=True)
    ev.record()
    t1 = time.perf_counter()
    loss.backward()
    t2 = time.perf_counter()
    ev.synchronize()
    t3 = time.perf_counter()
    opt.step()
    t4 = time.perf_counter()
    acc["fwd"] += t1 - t0; acc["bwd"] += t2 - t1; acc["wait"] += t3 - t2; acc["opt"] += t4 - t3
torch.cuda.synchronize()
tot = time.perf_counter() - t_all
print("wall %.2f ms/step | host: forward enqueue %.2f, backward enqueue %.2f, wait for loss %.2f, optimizer %.2f ms" % (
    1e3 * tot / N, *(1e3 * acc[k] / N for k in ("fwd", "bwd", "wait", "opt"))))
