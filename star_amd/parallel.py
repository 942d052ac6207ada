"""Scene-sharded data parallelism: one process per GPU, whole scenes per rank, one exchange per step --
the mean of the gradients over RCCL/xGMI.

The reference is single-process (nn.DataParallel commented out at completion/train_completion.py:120); this
is the multi-GPU step the north star adds.  Design for xGMI (point-to-point links, ring collectives are
per-link bound): the flat gradient buffer of star_amd.optim.FlatAdamW is laid out in backward-completion
order, so each finished stage is one contiguous range; consecutive stages are merged into buckets of at
least `bucket_mb` and all-reduced (AVG) asynchronously as soon as the backward chain has enqueued them, while
the remaining backward kernels keep the compute stream busy.  No collective is on the data path: scenes are
independent units (fusion only couples the agents inside one scene).

Two transports:
  RCCL ("nccl" backend, the product path): the collective is enqueued on RCCL's stream behind the compute stream and
      joined in finish() by a stream wait; the host never blocks.
  gloo with device tensors (rehearsal of the same control flow where RCCL cannot run: ranks sharing one GPU): every
      device<->host move is done HERE, on the compute stream, through ONE pinned staging buffer allocated up front;
      gloo only ever sees CPU tensors, from the calling thread, in finish().  (Round 1 handed device tensors to gloo, whose
      worker threads then allocate pinned memory, copy and synchronise on private streams while the backward thread is
      still launching -- the only multi-thread HIP use in the step -- and that run stalled intermittently with one rank
      inside a device synchronize)
"""
import torch
import torch.distributed as dist


class SceneParallel:
    def __init__(self, model, optimizer, process_group=None, bucket_mb=8.0, overlap=True, force=False):
        """overlap: hand each bucket over from the backward chain as soon as its gradients are enqueued (RCCL: the
        all-reduce itself; gloo rehearsal: the device-to-host staging copy); False = everything in finish().
        force: install the hooks even with one rank (tests of the RCCL call path on a single GPU)."""
        if not getattr(optimizer, "is_star_flat", False):
            raise TypeError("SceneParallel reduces the flat gradient buffer of star_amd.optim.FlatAdamW")
        self.model, self.opt, self.group = model, optimizer, process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(process_group) if dist.is_initialized() else 0
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
        # RCCL averages in the collective; gloo (rehearsal of the same logic) sums and divides afterwards
        self.native_avg = dist.is_initialized() and dist.get_backend(process_group) == "nccl"
        self.overlap = bool(overlap)
        self.deferred = []
        self.active = self.world > 1 or (force and dist.is_initialized())
        self._host = None           # gloo + device tensors: pinned staging image of the flat gradient buffer
        self._flag_stream = self._flag_event = None     # RCCL: side stream of the agreed non-finite-loss flag
        model._on_grads_ready = self._ready if self.active else None

    # ---- start-up -------------------------------------------------------------------------------------------------
    def _bcast(self, t, src):
        if self.native_avg or not t.is_cuda:
            dist.broadcast(t, src, group=self.group)
        else:                                  # gloo rehearsal: stage through the host ourselves
            h = t.detach().cpu()
            dist.broadcast(h, src, group=self.group)
            if self.rank != src:
                t.copy_(h)

    def broadcast_parameters(self, src=0):
        """Rank `src`'s state to every rank: the flat parameter buffer, every parameter that is NOT a view of it (frozen
        tables, VQ-STAR's EMA codebook and `_ema_w`, which no optimizer owns) and every buffer."""
        if not self.active:
            return
        self._bcast(self.opt.flat_p, src)
        flat = set(self.opt.offsets)
        with torch.no_grad():
            for name, prm in self.model.named_parameters():
                if name not in flat:
                    self._bcast(prm.data, src)
            for name, buf in self.model.named_buffers():
                self._bcast(buf, src)
        self.model._weights.epoch += 1

    # ---- per step --------------------------------------------------------------------------------------------------
    def _launch(self, rng):
        view = self.opt.flat_g[rng[0]:rng[1]]
        if self.native_avg:
            self.pending.append((dist.all_reduce(view, op=dist.ReduceOp.AVG, group=self.group, async_op=True), view, None))
        elif not view.is_cuda:                 # CPU tensors (unit tests of the bucket logic)
            self.pending.append((dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True), view, None))
        else:
            if self._host is None:
                self._host = torch.empty(self.opt.flat_g.numel(), dtype=torch.float32).pin_memory()
            host = self._host[rng[0]:rng[1]]
            host.copy_(view, non_blocking=True)                 # on the compute stream, behind the kernels that wrote `view`
            ev = torch.cuda.Event()
            ev.record()
            self.pending.append((None, view, (host, ev)))

    def _ready(self, stage):
        rng = self.close_at.get(stage)
        if rng is not None:
            if self.overlap:
                self._launch(rng)
            else:
                self.deferred.append(rng)

    def finish(self):
        """Join every outstanding bucket (call between backward and the optimizer step).  RCCL: stream waits only."""
        for rng in self.deferred:
            self._launch(rng)
        self.deferred = []
        for work, view, staged in self.pending:
            if staged is None:
                work.wait()
                if not self.native_avg:
                    view.div_(self.world)
            else:
                host, ev = staged
                ev.synchronize()
                dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)     # CPU tensor, calling thread
                host.div_(self.world)
                view.copy_(host, non_blocking=True)             # stream-ordered in front of the optimizer kernel
        self.pending = []

    def nonfinite_flag_to(self, loss, host_slot):
        """Queue "is the loss of ANY rank not finite" (a one-word MAX all-reduce over RCCL) and its copy into the pinned
        `host_slot` on a side stream; returns the event that marks the copy, or None when the transport cannot do that without
        blocking the host (gloo: use any_rank() on the host value instead).  The COMPUTE stream is not made to wait for the
        collective: the backward queued behind this call starts at once, and the flag's latency (and the ranks' skew at the end
        of the forward) stays off the step's critical path.  RCCL sees the flag before the step's gradient buckets on every rank."""
        if not (self.active and self.native_avg):
            return None
        if self._flag_stream is None:
            self._flag_stream = torch.cuda.Stream()
            self._flag_event = torch.cuda.Event()
        flag = (~torch.isfinite(loss.detach())).to(torch.float32).reshape(1)
        self._flag_stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self._flag_stream):
            dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=self.group)
            host_slot.copy_(flag, non_blocking=True)
            self._flag_event.record(self._flag_stream)
        flag.record_stream(self._flag_stream)
        return self._flag_event

    def any_rank(self, bad):
        """Host-side agreement (gloo rehearsal / CPU tests): True on every rank if `bad` is True on any."""
        if not self.active or self.native_avg:
            return bool(bad)
        t = torch.tensor([1.0 if bad else 0.0])
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return bool(t.item() > 0)
