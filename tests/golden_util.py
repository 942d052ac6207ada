"""Loading of the committed golden fixtures (tests/golden/*.npz, written by gen_golden.py)."""
import os

import numpy as np
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

MODEL_CASES = [
    "tiny_T1_comp_train", "tiny_T1_comp_infer", "tiny_T2_comp_train", "tiny_T3_comp_train",
    "tiny_T3_comp_infer", "tiny_T3_rand_train", "tiny_T3_rand_infer", "tiny_T3_partial_train",
    "tiny_T3_single_infer", "tiny_T2_notemp_train", "tiny_T2_rand_partial_train",
    "small_T2_comp_train", "small_T1_comp_infer",
]

# the same reference models after a few hundred AdamW steps on the structured law (gen_golden_trained.py): confident logits
TRAINED_CASES = [
    "trained_small_T2_comp_train", "trained_small_T2_comp_infer", "trained_small_T1_comp_infer", "trained_small_T1_comp_train",
    "trained_tiny_T3_rand_infer", "trained_tiny_T3_rand_train", "trained_tiny_T3_rand90_infer",
]

DIMS = {
    "tiny": dict(patch=8, chans=13, dim=32, depth=2, heads=2, ddim=32, ddepth=2, dheads=2),
    "small": dict(patch=8, chans=13, dim=64, depth=2, heads=2, ddim=48, ddepth=1, dheads=3),
}


def unpack(bits, shape, dtype=np.float32):
    n = int(np.prod(shape))
    return np.unpackbits(bits)[:n].reshape(tuple(int(s) for s in shape)).astype(dtype)


class Case:
    def __init__(self, name):
        z = np.load(os.path.join(GOLD, name + ".npz"))
        self.z, self.name = z, name
        self.B, self.A, self.T, self.S, rnd, train = (int(v) for v in z["meta"])
        self.mask = "random" if rnd else "complement"
        self.train = bool(train)
        self.mask_ratio = float(z["mask_ratio"])
        fl = z["flags"]
        self.flags = dict(encode_partial=bool(fl[0]), no_temp_emb=bool(fl[1]), decode_singletemp=bool(fl[2]))
        self.dims = dict(DIMS[name.replace("trained_", "").split("_")[0]])
        self.bev = torch.from_numpy(unpack(z["bev_bits"], z["bev_shape"]))
        self.next = torch.from_numpy(unpack(z["next_bits"], z["next_shape"])) if self.T > 1 else torch.zeros(self.A * self.B, 0)
        self.teacher = torch.from_numpy(unpack(z["teacher_bits"], z["teacher_shape"]))
        self.trans = z["trans"]
        self.num_agent = z["num_agent"]
        self.noise = torch.from_numpy(z["noise"][0])
        sd = np.load(os.path.join(GOLD, str(z["sd_name"]) + ".npz"))
        self.sd = {k: torch.from_numpy(sd[k]) for k in sd.files}
        self.loss = float(z["loss"])
        self.ind = unpack(z["ind_bits"], z["ind_shape"], np.int64)
        self.result = unpack(z["result_bits"], z["result_shape"])
        self.iou = z["iou"]

    def tap(self, k):
        return torch.from_numpy(self.z["tap_" + k])

    def grads(self):
        return {k[5:]: torch.from_numpy(self.z[k]) for k in self.z.files if k.startswith("grad_")}
