#!/usr/bin/env python3
"""How far ahead of the GPU is the host inside a training step?  Per step: host time spent enqueuing the forward, the backward and
the optimizer, and the time a device synchronize at the end of the step still has to wait (0 = the GPU had already run dry: the step
is host-bound at its end).  GPU box only."""
import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from star_amd.models import multiagent_mae as mm
from star_amd.optim import FlatAdamW
from star_amd.synthetic import make_batch
from star_amd.utils.CoModule import CoModule

dev = torch.device("cuda", 0)
B = int(os.environ.get("B", 8))
model = mm.amortized_ind_patch8_light(time_stamp=1, mask_method="complement").to(dev).set_compute_dtype("bf16")
opt = FlatAdamW(model, lr=1e-3, betas=(0.9, 0.95), weight_decay=0.05)
model.train()
cm = CoModule(model, opt, "ind_mae")
data = make_batch(B, 5, 1, seed=1, device=dev)
bev, nxt, teacher, trans, na = cm._unpack(data)
for _ in range(5):
    cm.step_mae_completion(data, B, 0.75, trainable=True)
torch.cuda.synchronize()
rows = []
for it in range(20):
    t0 = time.perf_counter()
    loss, result, _, ind = model(bev, nxt, teacher, trans, na, B, mask_ratio=0.75)
    t1 = time.perf_counter()
    loss.backward()
    t2 = time.perf_counter()
    opt.step()
    t3 = time.perf_counter()
    if os.environ.get("SYNC_EACH", "1") == "1":
        torch.cuda.synchronize()
    t4 = time.perf_counter()
    rows.append((t1 - t0, t2 - t1, t3 - t2, t4 - t3))
torch.cuda.synchronize()
r = rows[5:]
avg = [1e3 * sum(x[i] for x in r) / len(r) for i in range(4)]
print("B=%d  host enqueue: forward %.2f ms, backward %.2f ms, optimizer %.2f ms; wait in synchronize at the end %.2f ms; sum %.2f ms" % (B, *avg, sum(avg)))
# free-running (no synchronize between steps): wall time per step
t0 = time.perf_counter()
for it in range(20):
    loss, result, _, ind = model(bev, nxt, teacher, trans, na, B, mask_ratio=0.75)
    loss.backward()
    opt.step()
torch.cuda.synchronize()
print("free-running: %.2f ms per step" % (1e3 * (time.perf_counter() - t0) / 20))
