"""Move an optimizer's state tensors to a device after a CPU-side resume (star/utils/move_optim.py:3-15)."""
import torch


def optimizer_to(optim, device):
    for state in optim.state.values():
        for key, val in list(state.items()):
            if isinstance(val, torch.Tensor):
                state[key] = val.to(device)
