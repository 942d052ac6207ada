"""Scene-sharded data parallelism: one process per GPU, whole scenes per rank, one exchange per step --
the mean of the gradients over RCCL/xGMI.

The reference is single-process (nn.DataParallel commented out at completion/train_completion.py:120); this
is the multi-GPU step the north star adds.  Design for xGMI (point-to-point links, ring collectives are
per-link bound): the flat gradient buffer of star_amd.optim.FlatAdamW is laid out in backward-completion
order, so each finished stage is one contiguous range; consecutive stages are merged into buckets of at
least `bucket_mb` and all-reduced (AVG) asynchronously as soon as the backward chain has enqueued them, while
the remaining backward kernels keep the compute stream busy.  No collective is on the data path: scenes are
independent units (fusion only couples the agents inside one scene)."""
import torch
import torch.distributed as dist


class SceneParallel:
    def __init__(self, model, optimizer, process_group=None, bucket_mb=8.0, overlap=None, force=False):
        """overlap: launch each bucket's all-reduce from the backward chain as soon as its gradients are enqueued
        (default for RCCL, where the collective is stream-ordered); False = launch all buckets in finish().  The gloo
        rehearsal defaults to False: its GPU path stages through pinned host buffers allocated inside the call, which
        was seen to dead-lock against the kernel-launching backward thread when two ranks share one GPU.
        force: install the hooks even with one rank (tests of the RCCL call path on a single GPU)."""
        if not getattr(optimizer, "is_star_flat", False):
            raise TypeError("SceneParallel reduces the flat gradient buffer of star_amd.optim.FlatAdamW")
        self.model, self.opt, self.group = model, optimizer, process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        # merge stages into buckets: stage index at which a bucket closes -> [lo, hi)
        self.close_at, lo, acc = {}, None, 0
        limit = int(bucket_mb * (1 << 20) / 4)
        ranges = optimizer.stage_ranges
        for k, (a, b) in enumerate(ranges):
            lo = a if lo is None else lo
            acc += b - a
            if acc >= limit or k == len(ranges) - 1:
                self.close_at[k] = (lo, b)
                lo, acc = None, 0
        self.pending = []
        # RCCL averages in the collective; gloo (CPU rehearsal of the same logic) sums and divides afterwards
        self.native_avg = dist.is_initialized() and dist.get_backend(process_group) == "nccl"
        self.overlap = self.native_avg if overlap is None else bool(overlap)
        self.deferred = []
        self.active = self.world > 1 or (force and dist.is_initialized())
        model._on_grads_ready = self._ready if self.active else None

    def broadcast_parameters(self, src=0):
        if self.active:
            dist.broadcast(self.opt.flat_p, src, group=self.group)
            for name, buf in self.model.named_buffers():
                dist.broadcast(buf, src, group=self.group)
            self.model._weights.epoch += 1

    def _launch(self, rng):
        view = self.opt.flat_g[rng[0]:rng[1]]
        op = dist.ReduceOp.AVG if self.native_avg else dist.ReduceOp.SUM
        self.pending.append((dist.all_reduce(view, op=op, group=self.group, async_op=True), view))

    def _ready(self, stage):
        rng = self.close_at.get(stage)
        if rng is not None:
            if self.overlap:
                self._launch(rng)
            else:
                self.deferred.append(rng)

    def finish(self):
        """Make the compute stream wait for every outstanding bucket (call between backward and step)."""
        if self.deferred and not self.native_avg and self.opt.flat_g.is_cuda:
            # gloo stages device tensors through the host from its own threads; with the backward chain still running on a GPU
            # that the ranks share, its second step was seen to stall: the rehearsal path drains the device first
            torch.cuda.synchronize()
        for rng in self.deferred:
            self._launch(rng)
        self.deferred = []
        for work, view in self.pending:
            work.wait()
            if not self.native_avg:
                view.div_(self.world)
        self.pending = []
