#!/usr/bin/env python3
"""Inference latency per call, eager vs HIP-graph replay, for small batches.  usage: bench_infer_graph.py [scenes ...]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from star_amd.graph import GraphedInference  # noqa: E402
from star_amd.models import multiagent_mae  # noqa: E402
from star_amd.synthetic import make_batch  # noqa: E402
from star_amd.utils.CoModule import CoModule  # noqa: E402

DEV = "cuda"
model = multiagent_mae.amortized_ind_patch8_light(time_stamp=1, mask_method="complement").to(DEV).eval()
for B in [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8, 16]:
    data = make_batch(B, 5, 1, seed=1, device=DEV)
    bev, nxt, teacher, trans, na = CoModule._unpack(data)
    run = GraphedInference(model, B)

    def eager():
        with torch.no_grad():
            return model.inference(bev, nxt, teacher, trans, na, B)

    def graphed():
        return run(bev, nxt, teacher, trans, na)

    res = []
    for fn in (eager, graphed):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t = time.perf_counter()
        n = 30
        for _ in range(n):
            out = fn()
            out[0].item()                      # the caller reads the loss: one round trip per call, as test_completion.py does
        res.append((time.perf_counter() - t) / n * 1e3)
    print("scenes %2d (frames %3d): eager %7.3f ms/call  graph %7.3f ms/call  -> %6.0f vs %6.0f frames/s" % (
        B, 5 * B, res[0], res[1], 5 * B / res[0] * 1e3, 5 * B / res[1] * 1e3), flush=True)
