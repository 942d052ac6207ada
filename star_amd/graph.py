"""HIP-graph capture and replay of the inference path.

Every kernel of libstar_hip.so is launched on torch's current stream, the library allocates nothing and never
synchronises, so `model.inference(...)` can be captured whole into a HIP graph (torch.cuda.CUDAGraph) and replayed:

    run = GraphedInference(model, batch_size)
    loss, result, latent, ind = run(bev, bev_next, teacher, trans_matrices, num_agent)      # same tuple as model.inference

* inputs are copied into static buffers; the pose table is reduced on the host as always (ops.warp_thetas) and lands in a
  static device buffer through a pinned staging tensor, so the replay itself has no host dependency;
* the mask noise is drawn inside the graph from torch's graph-safe generator: every replay draws fresh noise;
* outputs are static tensors, overwritten by the next call (clone what must survive);
* the graph is re-captured when the input signature (shapes, strides, dtypes), the mask ratio or the weights change
  (parameter versions / the fused optimizer's epoch counter).
What it is for: callers that run under capture or must not touch the Python launch path per request.  It is NOT faster
here (tools/bench_infer_graph.py on MI355X: one 5-agent scene 1.30 ms eager vs 1.35 ms replayed, 16 scenes 10.3 vs 10.5 ms):
the ~190 dependent kernels of a small batch are bound by the GPU's kernel-to-kernel dispatch latency (~6 us each), which a
replayed graph pays as well, and the eager host enqueue already runs ahead of the GPU.
Training is not captured: the optimizer's step count and learning rate are host scalars of the AdamW launch."""
import torch

from . import ops


class GraphedInference:
    def __init__(self, model, batch_size, mask_ratio=0.75):
        self.model, self.batch_size, self.mask_ratio = model, batch_size, mask_ratio
        self.graph = None
        self.key = None
        self.static = None
        self.out = None
        self.theta = None
        self.theta_host = None
        self.theta_copied = None       # event: the last host-to-device copy out of theta_host has finished
        self.captures = 0

    def _weights_key(self):
        return (self.model._weights.epoch, sum(p._version for p in self.model.parameters()), self.model._compute_dtype)

    @staticmethod
    def _sig(t):
        return (tuple(t.shape), tuple(t.stride()), t.dtype, t.device)

    def _capture(self, tensors, theta_cpu):
        model = self.model
        dev = tensors[0].device
        # static copies that keep the caller's memory layout (the kernels read strided images as they are)
        self.static = [torch.empty_strided(t.shape, t.stride(), dtype=t.dtype, device=t.device).copy_(t) for t in tensors]
        self.theta_host = torch.empty(theta_cpu.shape, dtype=torch.float32).pin_memory()
        self.theta_host.copy_(theta_cpu)
        self.theta = self.theta_host.to(dev)
        self.theta_copied = torch.cuda.Event()
        bev, nxt, na = self.static
        teacher = bev                      # the path never reads the teacher frames (its loss target is the agent's own input)
        model._theta_override = self.theta
        try:
            with torch.no_grad():
                side = torch.cuda.Stream(device=dev)
                side.wait_stream(torch.cuda.current_stream(dev))
                with torch.cuda.stream(side):       # warm-up outside the capture: weight copies, kernel attributes, allocator
                    for _ in range(2):
                        model.inference(bev, nxt, teacher, None, na, self.batch_size, mask_ratio=self.mask_ratio)
                torch.cuda.current_stream(dev).wait_stream(side)
                self.graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.graph):
                    self.out = model.inference(bev, nxt, teacher, None, na, self.batch_size, mask_ratio=self.mask_ratio)
        finally:
            model._theta_override = None
        self.captures += 1

    def __call__(self, imgs1, imgs_next, teacher, trans_matrices, num_agent_tensor):
        if not imgs1.is_cuda:
            raise RuntimeError("GraphedInference needs its inputs on the GPU")
        if self.model.training:
            raise RuntimeError("GraphedInference replays model.inference in eval mode: call model.eval() first")
        dev = imgs1.device
        na = num_agent_tensor if num_agent_tensor.is_cuda else num_agent_tensor.to(dev)
        if na.dtype not in (torch.int32, torch.int64):
            na = na.long()
        nxt = imgs_next if imgs_next.is_cuda else imgs_next.to(dev)
        tensors = [imgs1, nxt, na]
        theta_cpu = ops.warp_thetas(trans_matrices)
        key = (tuple(self._sig(t) for t in tensors), tuple(theta_cpu.shape), self.mask_ratio, self._weights_key())
        if self.graph is None or key != self.key:
            self.graph = None
            self._capture(tensors, theta_cpu)
            self.key = key
        else:
            for dst, src in zip(self.static, tensors):
                dst.copy_(src, non_blocking=True)
            self.theta_copied.synchronize()        # the pinned staging tensor is free again
            self.theta_host.copy_(theta_cpu)
            self.theta.copy_(self.theta_host, non_blocking=True)
            self.theta_copied.record()
        self.graph.replay()
        return self.out
