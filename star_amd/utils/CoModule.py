"""Train / infer step wrapper for the STAR completion model -- the reference's `CoModule`
(star/utils/CoModule.py:13-41,91-156) for the `ind_mae` path: same constructor, attributes
(`model`, `optimizer`, `scheduler`, `mae_loss_scaler`) and method signatures / return tuples.
The vae / vqstar / late methods of the reference belong to other models and are not part of this path."""
import math
import sys

import torch

from .misc import NativeScalerWithGradNormCount as NativeScaler
from .move_optim import optimizer_to


class _HostTimes:
    """STAR_AMD_HOST_TIMES=1 (measurement): host wall time between the marks of the fused training step, printed at exit."""

    def __init__(self):
        import os
        self.on = os.environ.get("STAR_AMD_HOST_TIMES") == "1"
        self.t, self.acc, self.n = None, {}, 0
        if self.on:
            import atexit
            atexit.register(self.report)

    def mark(self, name):
        if not self.on:
            return
        import time
        now = time.perf_counter()
        if self.t is not None:
            self.acc[name] = self.acc.get(name, 0.0) + now - self.t
        self.t = now
        self.n += name == "opt"

    def report(self):
        import sys
        n = max(self.n, 1)
        print("host ms per step: " + "  ".join("%s %.3f" % (k, 1e3 * v / n) for k, v in self.acc.items()) +
              "   (fwd = from the previous step's optimizer launch to the end of this step's forward launches)", file=sys.stderr)


_HOST_T = _HostTimes()


class CoModule(object):
    def __init__(self, model, optimizer, com, parallel=None, graph_inference=False):
        """`parallel`: optional star_amd.parallel.SceneParallel (multi-GPU gradient averaging).
        `graph_inference`: infer_mae_completion / infer_vqstar_completion replay a captured HIP graph (star_amd.graph): same
        results (no speed-up measured: see that module); the returned tensors are overwritten by the next call."""
        self._graphed = {} if graph_inference else None
        self.mae_loss_scaler = NativeScaler()
        self.model = model
        self.optimizer = optimizer
        self.parallel = parallel
        self.scheduler = None            # only the 'late' / 'vqvae' baselines of the reference use one
        self._loss_host = None           # pinned scalar + event for the early loss read-back of the fused step
        self._loss_event = None
        if com not in ("ind_mae", "joint_mae", "vqstar"):
            raise NotImplementedError("star_amd accelerates the 'ind_mae' and 'vqstar' completion paths; com=%r is out of scope" % (com,))

    def resume_from_cpu(self, checkpoint, device, trainable=True):
        """Load a (reference-format) checkpoint that was read with map_location='cpu', then move to `device`."""
        self.model = self.model.cpu()
        self.model.load_state_dict(checkpoint["model_state_dict"])
        self.model = self.model.to(device)
        if trainable:
            self.optimizer.load_state_dict(checkpoint["optimizer_state_dict"])
            optimizer_to(self.optimizer, device)
            if checkpoint.get("mae_scaler_state_dict"):
                self.mae_loss_scaler.load_state_dict(checkpoint["mae_scaler_state_dict"])

    @staticmethod
    def _unpack(data):
        # NHWC -> NCHW as VIEWS (the kernels take strides): [A*B,1,H,W,C] -> [A*B,C,H,W]
        bev_seq = data['bev_seq'].squeeze(1).permute(0, 3, 1, 2)
        bev_seq_next = data['bev_seq_next']            # [A*B, T-1, H, W, C] or [A*B, 0]
        if bev_seq_next.dim() > 2:
            bev_seq_next = bev_seq_next.permute(0, 1, 4, 2, 3)
        bev_teacher = data['bev_seq_teacher'].squeeze(1).permute(0, 3, 1, 2)
        return bev_seq, bev_seq_next, bev_teacher, data['trans_matrices'], data['num_agent']

    def _step(self, loss, trainable):
        """Loss readback, finite check and parameter update shared by the mae and vqstar steps."""
        fused = trainable and getattr(self.optimizer, "is_star_flat", False)
        if fused:
            # The loss is copied to pinned host memory right behind the forward kernels, THEN the whole backward (and the
            # bucketed gradient all-reduce) is enqueued; the host waits for the copy only, so the finite check still precedes
            # the update while the GPU never idles on a round trip -- the optimizer launch is queued behind the backward.
            # Multi-rank: the ranks must take the stop/continue decision TOGETHER (a rank that exits alone leaves the others
            # blocked in the gradient all-reduce): a one-word MAX all-reduce of "my loss is not finite" is queued on RCCL's stream
            # ahead of the gradient buckets and lands in the pinned pair beside the loss through a side stream.
            if self._loss_host is None:
                self._loss_host = torch.zeros(2, dtype=torch.float32).pin_memory()
                self._loss_event = torch.cuda.Event()
            self._loss_host[1] = 0.0
            self._loss_host[0:1].copy_(loss.detach().reshape(1), non_blocking=True)
            self._loss_event.record()
            flag_event = self.parallel.nonfinite_flag_to(loss, self._loss_host[1:2]) if self.parallel is not None else None
            _HOST_T.mark("fwd")
            loss.backward()
            _HOST_T.mark("bwd")
            self._loss_event.synchronize()
            _HOST_T.mark("sync")
            if flag_event is not None:
                flag_event.synchronize()
            loss_value = float(self._loss_host[0])
            bad = not math.isfinite(loss_value) or float(self._loss_host[1]) > 0
            flag = flag_event
            if self.parallel is not None and flag is None:
                bad = self.parallel.any_rank(bad)                  # gloo rehearsal: agreed on the host
        else:
            loss_value = loss.item()
            bad = not math.isfinite(loss_value)
        if bad:
            print("Loss is {}, stopping training".format(loss_value))      # on every rank in the same step (reference: CoModule.py:116-122)
            sys.exit(1)
        if fused:
            if self.parallel is not None:
                self.parallel.finish()
            self.optimizer.step()
            _HOST_T.mark("opt")
        elif trainable:
            self.mae_loss_scaler(loss, self.optimizer, parameters=self.model.parameters(), update_grad=True)
            self.optimizer.zero_grad()
        return loss_value

    def step_mae_completion(self, data, batch_size, mask_ratio, loss_fn='mse', trainable=False):
        bev_seq, bev_seq_next, bev_teacher, trans_matrices, num_agent_tensor = self._unpack(data)
        loss, result, _, ind_pred = self.model(bev_seq, bev_seq_next, bev_teacher, trans_matrices, num_agent_tensor,
                                               batch_size, mask_ratio=mask_ratio)
        return self._step(loss, trainable), result, ind_pred.detach()

    def step_vqstar_completion(self, data, batch_size, mask_ratio, loss_fn='mse', trainable=False):
        """star/utils/CoModule.py:193-229 -> (loss, result, ind_pred, perplexity)."""
        bev_seq, bev_seq_next, bev_teacher, trans_matrices, num_agent_tensor = self._unpack(data)
        loss, result, _, ind_pred, perplexity = self.model(bev_seq, bev_seq_next, bev_teacher, trans_matrices, num_agent_tensor,
                                                           batch_size, mask_ratio=mask_ratio)
        return self._step(loss, trainable), result, ind_pred.detach(), perplexity.detach()

    def infer_vqstar_completion(self, data, batch_size, mask_ratio, loss_fn='mse'):
        """star/utils/CoModule.py:231-253 -> (loss, result, ind_pred, perplexity, encodings)."""
        bev_seq, bev_seq_next, bev_teacher, trans_matrices, num_agent_tensor = self._unpack(data)
        with torch.no_grad():
            loss, result, ind_pred, perplexity, encodings = self._inference(bev_seq, bev_seq_next, bev_teacher, trans_matrices,
                                                                            num_agent_tensor, batch_size, mask_ratio)
        return loss.item(), result, ind_pred.detach(), perplexity.detach(), encodings.detach()

    def infer_mae_completion(self, data, batch_size, mask_ratio, loss_fn='mse'):
        bev_seq, bev_seq_next, bev_teacher, trans_matrices, num_agent_tensor = self._unpack(data)
        with torch.no_grad():
            loss, result, latent, ind_pred = self._inference(bev_seq, bev_seq_next, bev_teacher, trans_matrices, num_agent_tensor,
                                                             batch_size, mask_ratio)
        return loss.item(), result, ind_pred.detach(), latent.detach()

    def _inference(self, bev_seq, bev_seq_next, bev_teacher, trans_matrices, num_agent_tensor, batch_size, mask_ratio):
        if self._graphed is None:
            return self.model.inference(bev_seq, bev_seq_next, bev_teacher, trans_matrices, num_agent_tensor, batch_size, mask_ratio=mask_ratio)
        run = self._graphed.get((batch_size, mask_ratio))
        if run is None:
            from star_amd.graph import GraphedInference
            run = self._graphed[(batch_size, mask_ratio)] = GraphedInference(self.model, batch_size, mask_ratio)
        return run(bev_seq, bev_seq_next, bev_teacher, trans_matrices, num_agent_tensor)
