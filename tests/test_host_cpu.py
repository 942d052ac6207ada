"""CPU-side checks of the host package against the golden fixtures (no GPU, no HIP call)."""
import types
from functools import partial

import numpy as np
import torch
import torch.nn as nn

from golden_util import GOLD
from star_amd.models import multiagent_mae as mm
from star_amd.utils import lr_sched
from star_amd.utils.pos_embed import get_2d_sincos_pos_embed


def test_pos_embed_bits():
    z = np.load(GOLD + "/standalone.npz")
    for key in ("pos_384_32", "pos_256_32", "pos_32_32", "pos_768_16", "pos_48_4"):
        _, d, g = key.split("_")
        assert np.array_equal(get_2d_sincos_pos_embed(int(d), int(g), cls_token=True).astype(np.float32), z[key]), key


def test_lr_schedule_table():
    z = np.load(GOLD + "/standalone.npz")
    args = types.SimpleNamespace(lr=1e-3, min_lr=1e-5, warmup_epochs=4, nepoch=30)
    opt = types.SimpleNamespace(param_groups=[{"lr": 0.0}, {"lr": 0.0, "lr_scale": 0.5}])
    got = np.array([lr_sched.adjust_learning_rate(opt, float(e), args) for e in z["lr_epochs"]])
    assert np.array_equal(got, z["lr_values"])
    assert opt.param_groups[1]["lr"] == 0.5 * opt.param_groups[0]["lr"]


def test_factory_inventory_matches_reference():
    z = np.load(GOLD + "/standalone.npz")
    inv = dict(s.split("|", 1) for s in z["inventory"] if "nparams" not in s)
    for fn, want in inv.items():
        m = getattr(mm, fn)(time_stamp=2, mask_method="complement")
        mine = ";".join("%s:%s" % (k, "x".join(map(str, v.shape))) for k, v in m.state_dict().items())
        assert mine == want, fn


def test_seeded_construction_draws_the_reference_weights():
    """Same seed -> same initial weights as the reference constructor (parameter creation order,
    initialisers and the frozen tables all match)."""
    for sd_name, T, dims in (("sd_tiny_T3", 3, dict(embed_dim=32, depth=2, num_heads=2, decoder_embed_dim=32, decoder_depth=2, decoder_num_heads=2)),
                             ("sd_small_T2", 2, dict(embed_dim=64, depth=2, num_heads=2, decoder_embed_dim=48, decoder_depth=1, decoder_num_heads=3))):
        torch.manual_seed(0)
        m = mm.AmortizedIndivMMAEViT(img_size=32 if "tiny" in sd_name else 64, patch_size=8, in_chans=13, mlp_ratio=4,
                                     norm_layer=partial(nn.LayerNorm, eps=1e-6), time_stamp=T, mask_method="complement", **dims)
        torch.manual_seed(7)   # the same perturbation gen_golden.py applies after construction
        with torch.no_grad():
            m.decoder_temp_embed.normal_(std=0.02)
            for n_, p in m.named_parameters():
                if n_.endswith(".bias") or "norm" in n_:
                    p.add_(torch.randn_like(p) * 0.05)
        ref = np.load(GOLD + "/" + sd_name + ".npz")
        for k, v in m.state_dict().items():
            assert np.array_equal(v.numpy(), ref[k]), (sd_name, k)


def test_out_of_scope_constructors_say_so():
    import pytest
    with pytest.raises(NotImplementedError):
        mm.joint_bev_mae_vit_base_patch8(time_stamp=1)
    with pytest.raises(NotImplementedError):
        mm.amortized_ind_patch8_light(time_stamp=1, mask_method="stripes")


def test_tap_and_conv_words_match_the_header():
    """ops.wgrad_tap / the conv word of ops.gemm are the STAR_WGRAD_TAP / STAR_GEMM_CONV3X3 macros of include/star_hip.h, and the ctypes
    structs have the header's sizes (a field added to one side only would shift every argument behind it)."""
    import ctypes as C
    import os
    import re
    import subprocess
    import tempfile
    from star_amd import _lib as L
    from star_amd import ops
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = r'''
#include <stdio.h>
#include "star_hip.h"
int main(void) {
    printf("%d %d %d %d %zu %zu\n", STAR_WGRAD_TAP(-1, 1, 6, 7), STAR_WGRAD_TAP(0, 0, 8, 8), STAR_GEMM_CONV3X3(6, 7), STAR_GEMM_CONV3X3(8, 8),
           sizeof(StarGemm), sizeof(StarWgradJob));
    return 0;
}
'''
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.c"), "w").write(src)
        subprocess.check_call(["gcc", "-I", os.path.join(root, "include"), os.path.join(d, "t.c"), "-o", os.path.join(d, "t")])
        out = subprocess.check_output([os.path.join(d, "t")]).decode().split()
    t1, t2, c1, c2, sg, sj = (int(x) for x in out)
    assert ops.wgrad_tap(-1, 1, 128, 64) == t1 and ops.wgrad_tap(0, 0, 256, 256) == t2
    assert (0x40000000 | 6 | (7 << 5)) == c1 and (0x40000000 | 8 | (8 << 5)) == c2
    assert C.sizeof(L.StarGemm) == sg and C.sizeof(L.StarWgradJob) == sj
