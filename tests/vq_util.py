"""Loading of the VQ-STAR golden fixtures (tests/golden/vq_*.npz, written by gen_golden_vq.py)."""
import os

import numpy as np
import torch

from golden_util import GOLD, unpack

VQ_CASES = ["vq_T1_comp_train", "vq_T3_comp_infer", "vq_T2_rand_train"]
VQ_EMA_CASES = ["vqema_T2_comp_train", "vqema_T1_comp_infer"]
VQ_DIMS = dict(patch=8, chans=13, dim=32, depth=2, heads=2, ddim=32, ddepth=2, dheads=2)


class VQCase:
    def __init__(self, name):
        z = np.load(os.path.join(GOLD, name + ".npz"))
        self.z, self.name = z, name
        self.B, self.A, self.T, self.S, rnd, train, self.K, self.E = (int(v) for v in z["meta"])
        self.mask = "random" if rnd else "complement"
        self.train = bool(train)
        self.mask_ratio, self.cc = float(z["mask_ratio"]), float(z["cc"])
        self.decay = float(z["decay"]) if "decay" in z.files else 0.0
        self.bev = torch.from_numpy(unpack(z["bev_bits"], z["bev_shape"]))
        self.next = torch.from_numpy(unpack(z["next_bits"], z["next_shape"])) if self.T > 1 else torch.zeros(self.A * self.B, 0)
        self.teacher = torch.from_numpy(unpack(z["teacher_bits"], z["teacher_shape"]))
        self.trans, self.num_agent = z["trans"], z["num_agent"]
        self.noise = torch.from_numpy(z["noise"][0])
        sd = np.load(os.path.join(GOLD, "sd_" + name + ".npz"))
        self.sd = {k: torch.from_numpy(sd[k]) for k in sd.files}
        self.loss, self.perplexity = float(z["loss"]), float(z["perplexity"])
        self.ind = unpack(z["ind_bits"], z["ind_shape"], np.int64)
        self.result = unpack(z["result_bits"], z["result_shape"])

    def tap(self, k):
        return torch.from_numpy(self.z["tap_" + k])

    def grads(self):
        return {k[5:]: torch.from_numpy(self.z[k]) for k in self.z.files if k.startswith("grad_")}
