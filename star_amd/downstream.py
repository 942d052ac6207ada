"""Hand-off of the completed BEV to the downstream detection / segmentation heads (BASELINE config 5: frozen STAR feeding
coperception's FaFNet / UNet, which are third-party and out of scope).  The reference does the hand-off inline in its
evaluation scripts; these helpers restate exactly those lines so that a caller of the batched `inference()` gets the tensors
in the layouts the heads take:

  detection     completion/test_codet.py:263-275   infer_mae_completion(data, batch_size=1, ...) -> result [A, 13, 256, 256]
                -> unsqueeze(1).permute(0, 1, 3, 4, 2) = [A, 1, 256, 256, 13], which REPLACES data["bev_seq"]
  segmentation  completion/test_coseg.py:262-286   model.inference(bev [A*B,13,256,256], next, teacher, trans, num_agent, B)
                -> result.permute(0, 2, 3, 1) = [A*B, 256, 256, 13] -> data["bev_seq"]

Both are views of the fused float32 map (values in {0, 1}); `contiguous=True` materialises the channels-last tensor once, for
heads that start with .view() / .reshape() on it."""
import torch


def det_input(result, contiguous=False):
    """[N, C, H, W] fused map -> [N, 1, H, W, C]  (test_codet.py:271-272)."""
    out = result.unsqueeze(1).permute(0, 1, 3, 4, 2)
    return out.contiguous() if contiguous else out


def seg_input(result, contiguous=False):
    """[N, C, H, W] fused map -> [N, H, W, C]  (test_coseg.py:281)."""
    out = result.permute(0, 2, 3, 1)
    return out.contiguous() if contiguous else out


def complete_for_detection(co_module, data, batch_size=1, mask_ratio=0.75, contiguous=False):
    """test_codet.py:263-275: run the frozen completion model on `data` (the per-batch dict of the completion scripts) and
    substitute the completed map for data["bev_seq"].  Returns (data, loss)."""
    loss, completed, _, _ = co_module.infer_mae_completion(data, batch_size=batch_size, mask_ratio=mask_ratio)
    data["bev_seq"] = det_input(completed, contiguous)
    return data, loss


def complete_for_segmentation(model, data, bev_seq, bev_seq_next, bev_teacher, trans_matrices, num_agent_tensor, batch_size, mask_ratio=0.75,
                              contiguous=False):
    """test_coseg.py:262-286: the segmentation script calls model.inference directly (no CoModule) with channel-first inputs
    and hands the heads the channels-last map in data["bev_seq"]."""
    with torch.no_grad():
        _, completed, _, _ = model.inference(bev_seq, bev_seq_next, bev_teacher, trans_matrices, num_agent_tensor, batch_size, mask_ratio=mask_ratio)
    data["bev_seq"] = seg_input(completed, contiguous)
    return data
