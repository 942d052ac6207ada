#!/usr/bin/env python3
"""Probe: does running two half-batch forward+backward chains on two HIP streams beat one full-batch chain?
Two model copies (own flat gradient buffers) x 4 scenes each on their own stream vs one model x 8 scenes.  No optimizer step.
usage: python tools/probe_two_streams.py [iters]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from star_amd.models import multiagent_mae  # noqa: E402
from star_amd.optim import FlatAdamW  # noqa: E402
from star_amd.synthetic import make_batch  # noqa: E402
from star_amd.utils.CoModule import CoModule  # noqa: E402

DEV = "cuda"
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10


def mk():
    m = multiagent_mae.amortized_ind_patch8_light(time_stamp=1, mask_method="complement").to(DEV)
    opt = FlatAdamW(m, lr=1e-3)
    m.train()
    return m, CoModule(m, opt, "ind_mae")


def fb(m, cm, data, b):
    args = cm._unpack(data)
    loss = m(*args, b, mask_ratio=0.75)[0]
    loss.backward()


m8, c8 = mk()
mA, cA = mk()
mB, cB = mk()
d8 = make_batch(8, 5, 1, seed=1, device=DEV)
dA = make_batch(4, 5, 1, seed=2, device=DEV)
dB = make_batch(4, 5, 1, seed=3, device=DEV)
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()


def run_one():
    fb(m8, c8, d8, 8)


def run_serial():
    fb(mA, cA, dA, 4)
    fb(mB, cB, dB, 4)


def run_two():
    cur = torch.cuda.current_stream()
    sA.wait_stream(cur)
    sB.wait_stream(cur)
    with torch.cuda.stream(sA):
        fb(mA, cA, dA, 4)
    with torch.cuda.stream(sB):
        fb(mB, cB, dB, 4)
    cur.wait_stream(sA)
    cur.wait_stream(sB)


for name, fn in (("one chain, 8 scenes", run_one), ("two chains serial, 4+4 scenes", run_serial), ("two chains on two streams, 4+4", run_two)):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(iters):
        fn()
    host = time.perf_counter() - t
    torch.cuda.synchronize()
    tot = time.perf_counter() - t
    print("%-36s %7.3f ms per fwd+bwd of 8 scenes (host enqueue %6.3f ms)" % (name, tot / iters * 1e3, host / iters * 1e3), flush=True)
