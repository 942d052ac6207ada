#!/usr/bin/env python3
"""Is the training step launch-bound?  Host time to ENQUEUE a step (no sync) vs GPU time per step."""
import os, sys, time, types
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
t0 = time.perf_counter(); host = 0.0
for _ in range(N):
    a = time.perf_counter()
    cm.step_mae_completion(data, 8, 0.75, trainable=True)
    host += time.perf_counter() - a
torch.cuda.synchronize()
tot = time.perf_counter() - t0
print("per step: wall %.2f ms, host time inside step() %.2f ms (includes the loss.item() sync)" % (1e3 * tot / N, 1e3 * host / N))
# host-only cost: run the step under a profiler of python time with GPU work removed is not possible; instead time with an empty GPU queue:
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(5):
    cm.step_mae_completion(data, 8, 0.75, trainable=True)
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(14)
