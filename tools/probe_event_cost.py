#!/usr/bin/env python3
"""What does an event record between two dependent kernels cost the stream it is recorded on?  A chain of N medium kernels, plain,
with a torch event recorded after every kernel, and with a second stream waiting on the chain after every kernel (wait_stream).
GPU box only."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from star_amd import ops

dev = "cuda"
M, D, N = 40960, 384, 200
x = torch.randn(M, D, device=dev)
gam, bet = torch.ones(D, device=dev), torch.zeros(D, device=dev)
side = torch.cuda.Stream()
evs = [torch.cuda.Event() for _ in range(N)]
import ctypes
hip = ctypes.CDLL("libamdhip64.so")
hip.hipEventCreateWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint]
hip.hipEventRecord.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
hip.hipStreamWaitEvent.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint]


def raw_events(flags):
    out = []
    for _ in range(N):
        e = ctypes.c_void_p()
        assert hip.hipEventCreateWithFlags(ctypes.byref(e), flags) == 0
        out.append(e)
    return out


RAW = {"raw_dev": raw_events(0x2 | 0x40000000), "raw_nofence": raw_events(0x2 | 0x20000000), "raw_default": raw_events(0x2)}


def chain(mode):
    main = torch.cuda.current_stream()
    for i in range(N):
        ops.layernorm_fwd(x, gam, bet, 1e-6, 0, want_stats=False)
        if mode == "event":
            evs[i].record()
        elif mode == "wait_stream":
            side.wait_stream(main)
        elif mode in RAW:
            assert hip.hipEventRecord(RAW[mode][i], ctypes.c_void_p(main.cuda_stream)) == 0
            assert hip.hipStreamWaitEvent(ctypes.c_void_p(side.cuda_stream), RAW[mode][i], 0) == 0


for mode in ("plain", "event", "wait_stream", "raw_default", "raw_dev", "raw_nofence", "plain"):
    chain(mode)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        chain(mode)
    e1.record()
    torch.cuda.synchronize()
    print("%-12s %.2f us per kernel" % (mode, e0.elapsed_time(e1) * 1e3 / (5 * N)))
