"""Loading of the joint-model fixtures (tests/golden/joint*.npz, written by gen_golden_joint.py)."""
import os

import numpy as np
import torch

from golden_util import GOLD, unpack

JOINT_CASES = ["joint_T1_comp_train", "joint_T2_rand_train", "joint_T2_comp_eval", "joint16_T3_comp_train"]


class JointCase:
    def __init__(self, name):
        z = np.load(os.path.join(GOLD, name + ".npz"))
        self.z, self.name = z, name
        self.B, self.A, self.T, self.S, self.patch, rnd, train = (int(v) for v in z["meta"])
        self.mask = "random" if rnd else "complement"
        self.train = bool(train)
        self.mask_ratio = float(z["mask_ratio"])
        self.bev = torch.from_numpy(unpack(z["bev_bits"], z["bev_shape"]))
        self.next = torch.from_numpy(unpack(z["next_bits"], z["next_shape"])) if self.T > 1 else torch.zeros(self.A * self.B, 0)
        self.teacher = torch.from_numpy(unpack(z["teacher_bits"], z["teacher_shape"]))
        self.trans, self.num_agent = z["trans"], z["num_agent"]
        self.noise = torch.from_numpy(z["noise"])
        self.sd = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd_")}
        self.after = {k[6:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("after_")}
        self.loss = float(z["loss"])
        self.result = z["result"].astype(np.int64)
        self.pred = torch.from_numpy(z["pred"])
        self.dims = dict(patch=self.patch, chans=13, dim=32, depth=1, heads=2, ddim=64, ddepth=1, dheads=2)

    def tap(self, k):
        return torch.from_numpy(self.z["tap_" + k])

    def grads(self):
        return {k[5:]: torch.from_numpy(self.z[k]) for k in self.z.files if k.startswith("grad_")}
