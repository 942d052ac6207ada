"""Loss scaler with gradient-norm readout, mirroring star/utils/misc.py:251-292
(`NativeScalerWithGradNormCount`, `get_grad_norm_`): scale -> backward -> unscale -> norm -> step -> update.
The path computes in fp32/bf16 (same exponent range), so the scale is exact powers of two and never
triggers; it is kept because the reference checkpoints store the scaler state."""
import math

import torch

inf = math.inf


def get_grad_norm_(parameters, norm_type: float = 2.0) -> torch.Tensor:
    if isinstance(parameters, torch.Tensor):
        parameters = [parameters]
    grads = [p.grad.detach() for p in parameters if p.grad is not None]
    if not grads:
        return torch.tensor(0.)
    if float(norm_type) == inf:
        return max(g.abs().max() for g in grads)
    return torch.norm(torch.stack(torch._foreach_norm(grads, float(norm_type))), float(norm_type))


class NativeScalerWithGradNormCount:
    state_dict_key = "amp_scaler"

    def __init__(self):
        self._scaler = torch.amp.GradScaler("cuda") if torch.cuda.is_available() else torch.amp.GradScaler("cpu", enabled=False)

    def __call__(self, loss, optimizer, clip_grad=None, parameters=None, create_graph=False, update_grad=True):
        self._scaler.scale(loss).backward(create_graph=create_graph)
        if not update_grad:
            return None
        self._scaler.unscale_(optimizer)
        if clip_grad is not None:
            assert parameters is not None
            norm = torch.nn.utils.clip_grad_norm_(parameters, clip_grad)
        else:
            norm = get_grad_norm_(parameters)
        self._scaler.step(optimizer)
        self._scaler.update()
        return norm

    def state_dict(self):
        return self._scaler.state_dict()

    def load_state_dict(self, state_dict):
        self._scaler.load_state_dict(state_dict)
