#!/usr/bin/env python3
"""Read the in-kernel s_memtime stamps of the forward attention loop (variant library built with -DATT_STAMP)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from star_amd import ops, _lib
g = torch.Generator().manual_seed(0)
NB, L, H, hd = 40, 1024, 12, 32
qkv = torch.randn(NB * L, 3 * H * hd, generator=g).cuda().bfloat16()
for _ in range(3):
    ops.attn_fwd(qkv, NB, L, H, hd)
torch.cuda.synchronize()
buf = (C.c_ulonglong * 128)()
h = C.CDLL(_lib.LIB_PATH)
h.star_debug_stamps(buf)
v = list(buf)[:16]
names = ["top", "after wait+barrier", "after DMA issue", "after S mfma + tr issue", "after max", "after exp", "after tr_wait", "after PV mfma"]
for itn in range(2):
    r = v[itn * 8:(itn + 1) * 8]
    print("iteration", 8 + itn, " total", (v[(itn + 1) * 8] - r[0]) if itn == 0 else (r[7] - r[0]))
    for k in range(1, 8):
        print("   %-26s +%5d" % (names[k], r[k] - r[k - 1]))
