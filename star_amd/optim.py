"""Fused AdamW over one flat parameter buffer (star_adamw) -- the optimiser half of the training step.

Reference behaviour kept (completion/train_completion.py:124-125 + timm 0.3.2 add_weight_decay):
AdamW(lr, betas=(0.9, 0.95), eps=1e-8), decoupled weight decay on tensors with >= 2 dims that are not
biases, none on 1-D tensors / biases; parameters that never receive a gradient (cls_token, temp_embed, ...)
are never touched.  The reference steps ~150 tensors one by one; here the parameters are re-homed as views
of ONE float32 buffer laid out in backward-completion order, the backward kernels write their gradients
straight into a second flat buffer, and the whole update is one launch.  The same flat gradient buffer is
what star_amd.parallel reduces over RCCL, bucket by bucket, while backward is still running.

Gradients are OVERWRITTEN by each backward (no accumulation across micro-batches)."""
import torch

from . import engine, ops


def add_weight_decay(model, weight_decay=1e-5, skip_list=()):
    """timm 0.3.2 optim_factory.add_weight_decay (used at train_completion.py:124)."""
    decay, no_decay = [], []
    for name, param in model.named_parameters():
        if not param.requires_grad:
            continue
        if len(param.shape) == 1 or name.endswith(".bias") or name in skip_list:
            no_decay.append(param)
        else:
            decay.append(param)
    return [{'params': no_decay, 'weight_decay': 0.}, {'params': decay, 'weight_decay': weight_decay}]


class FlatAdamW:
    is_star_flat = True

    def __init__(self, model, lr=1e-3, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.05, grouping="timm"):
        """grouping: layout of state_dict()'s param groups -- "timm" = add_weight_decay's two groups (the ind_mae
        optimizer, train_completion.py:124-125), "flat" = one group over model.parameters() (the vqstar optimizer,
        optim.Adam(model.parameters()) at train_completion.py:128; use weight_decay=0 with it)."""
        assert grouping in ("timm", "flat"), grouping
        self.grouping = grouping
        self.model = model
        cfg = model._cfg()
        named = dict(model.named_parameters())
        self.stages = [[n for n in st if named[n].requires_grad] for st in engine.stages(cfg)]
        self.names = [n for st in self.stages for n in st]
        dev = named[self.names[0]].device
        if dev.type != "cuda":
            raise RuntimeError("FlatAdamW needs the model on the GPU first (model.to('cuda'))")
        sizes = [named[n].numel() for n in self.names]
        pad = lambda v: (v + 3) // 4 * 4    # noqa: E731  keep every tensor 16-byte aligned for the vector kernels
        self.offsets, off = {}, 0
        for n, sz in zip(self.names, sizes):
            self.offsets[n] = off
            off += pad(sz)
        self.total = off
        self.flat_p = torch.zeros(off, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(off, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(off, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(off, dtype=torch.float32, device=dev)
        mask = torch.zeros(off, dtype=torch.uint8, device=dev)
        views = {}
        for n, sz in zip(self.names, sizes):
            prm, o = named[n], self.offsets[n]
            self.flat_p[o:o + sz].copy_(prm.detach().reshape(-1))
            prm.data = self.flat_p[o:o + sz].view(prm.shape)
            views[n] = self.flat_g[o:o + sz].view(prm.shape)
            prm.grad = views[n]
            if not (prm.dim() == 1 or n.endswith(".bias")):
                mask[o:o + sz] = 1
        self.wd_mask = mask
        model._grad_views = views
        self.stage_ranges = []          # [lo, hi) of every stage in the flat buffers
        for st in self.stages:
            lo = self.offsets[st[0]]
            last = st[-1]
            self.stage_ranges.append((lo, self.offsets[last] + pad(named[last].numel())))
        self.param_groups = [dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, params=[named[n] for n in self.names])]
        self.step_count = 0

    # ---- torch.optim-like surface ----------------------------------------------------------------------
    def zero_grad(self, set_to_none=False):
        pass    # every backward overwrites the flat gradient buffer

    def step(self):
        g = self.param_groups[0]
        self.step_count += 1
        ops.adamw(self.flat_p, self.flat_g, self.exp_avg, self.exp_avg_sq, self.wd_mask, g["lr"], g["betas"][0], g["betas"][1],
                  g["eps"], g["weight_decay"], self.step_count, gscale=self.model._grad_scale)
        self.model._weights.epoch += 1          # low-precision weight copies are stale now

    def grad_norm(self):
        """L2 norm of the (unscaled) gradient, as the reference's get_grad_norm_ reads it; a device scalar."""
        return ops.sumsq(self.flat_g).sqrt()

    def _groups(self):
        wd = self.param_groups[0]["weight_decay"]
        if self.grouping == "flat":
            return [{"params": list(self.model.parameters()), "weight_decay": wd}]
        return add_weight_decay(self.model, wd)

    def state_dict(self):
        """torch.optim.AdamW layout over timm's two groups (no-decay first), so checkpoints interchange with
        the reference's `optimizer_state_dict`."""
        order = self._groups()
        ids = {id(p): i for i, p in enumerate([p for grp in order for p in grp["params"]])}
        named = dict(self.model.named_parameters())
        state = {}
        if self.step_count > 0:
            for n in self.names:
                o, sz = self.offsets[n], named[n].numel()
                state[ids[id(named[n])]] = dict(step=torch.tensor(float(self.step_count)),
                                                exp_avg=self.exp_avg[o:o + sz].view(named[n].shape).clone(),
                                                exp_avg_sq=self.exp_avg_sq[o:o + sz].view(named[n].shape).clone())
        g = self.param_groups[0]
        groups, base = [], 0
        for grp in order:
            groups.append(dict(lr=g["lr"], betas=g["betas"], eps=g["eps"], weight_decay=grp["weight_decay"], amsgrad=False,
                               params=list(range(base, base + len(grp["params"])))))
            base += len(grp["params"])
        return dict(state=state, param_groups=groups)

    def load_state_dict(self, sd):
        plist = [p for grp in self._groups() for p in grp["params"]]
        names = {id(p): n for n, p in self.model.named_parameters()}
        for idx, st in sd["state"].items():
            n = names[id(plist[int(idx)])]
            if n not in self.offsets:
                continue
            o, sz = self.offsets[n], plist[int(idx)].numel()
            self.exp_avg[o:o + sz].copy_(st["exp_avg"].reshape(-1))
            self.exp_avg_sq[o:o + sz].copy_(st["exp_avg_sq"].reshape(-1))
            self.step_count = int(float(st["step"]))
        if sd.get("param_groups"):
            self.param_groups[0]["lr"] = sd["param_groups"][0]["lr"]
