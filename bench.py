#!/usr/bin/env python3
"""Headline benchmark: STAR completion TRAINING step (forward + backward + AdamW), batch 8 scenes x 5 agents of
synthetic 256x256x13 BEV per GPU, model amortized_ind_patch8_light, bf16 MFMA compute
(BASELINE.json configs[1]; with --gpus N it is configs[2]: scene-sharded data parallel, RCCL gradient all-reduce).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Prints ONE JSON line on rank 0: BEV frames/s of the whole job (a frame = one agent's completed 256x256x13 map),
plus `roofline` (dominant kernel, timed with HIP events inside the timed region) and, at N=1, `cpu_baseline`
(the CPU oracle's training step timed on this box's host cores on a bounded sample)."""
import argparse
import json
import os
import sys
import time
import types

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0      # dense bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=8, help="scenes per GPU")
    ap.add_argument("--agents", type=int, default=5)
    ap.add_argument("--time_stamp", type=int, default=1)
    ap.add_argument("--mask", default="complement", choices=["complement", "random"])
    ap.add_argument("--mask_ratio", type=float, default=0.75)
    ap.add_argument("--model", default="amortized_ind_patch8_light")
    ap.add_argument("--com", default="ind_mae", choices=["ind_mae", "vqstar"], help="vqstar: the VQ-STAR variant (Makefile sizes K=8192, E=32)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--mode", default="train", choices=["train", "infer"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile-all", action="store_true", help="print the per-kernel-family time table to stderr")
    return ap.parse_args()


# kernel family (ops.Profiler tag) -> kernel-name prefixes in the rocprofv3 PMC summary committed under profiles/
_FAMILY_KERNELS = {"gemm_nt": ("void gemm_nt_fast_kernel", "void gemm_nt_wreg_kernel", "void gemm_nt_kernel"), "gemm_wgrad": ("void gemm_tn_", "wgrad_reduce_kernel"),
                   "attn_fwd": ("void attn_fwd_",), "attn_bwd": ("void attn_bwd_", "void attn_delta_"),
                   "layernorm_fwd": ("void ln_fwd_",), "layernorm_bwd": ("void ln_bwd_", "ln_bwd_reduce"),
                   "ce_argmax": ("void ce_argmax_",), "warp_fuse": ("warp_fuse_", "wf_pack_bits")}


def _pmc_traffic(family):
    """HBM-side bytes per launch of the family's kernels from the committed PMC summary (tools/pmc_traffic.sh: separate
    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this same bench command, gfx950 FETCH_SIZE correction applied).
    Counters cannot be collected inside the timed run, so this is read from profiles/; None if the file is absent."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    if not os.path.exists(path) or family not in _FAMILY_KERNELS:
        return None
    with open(path) as f:
        kernels = json.load(f)["kernels"]
    tot, n = 0.0, 0
    for name, v in kernels.items():
        if name.startswith(_FAMILY_KERNELS[family]):
            tot += (v["fetch_bytes_per_launch"] + v["write_bytes_per_launch"]) * v["launches"]
            n += v["launches"] if not name.startswith(("wgrad_reduce", "ln_bwd_reduce", "void attn_delta_", "wf_pack_bits")) else 0
    return round(tot / n) if n else None


def _pmc_mfma_util(family):
    """Fraction of SIMD cycles with the MFMA pipe busy for the family, from the committed PMC summary (tools/pmc_mfma.sh over
    this same bench command); None if absent."""
    path = os.path.join(ROOT, "profiles", "r01_l_pmc_mfma.json")
    if not os.path.exists(path):
        return None
    with open(path) as f:
        v = json.load(f)["families"].get(family)
    return round(v["mfma_util_busy_cycles"], 4) if v and v.get("mfma_util_busy_cycles") is not None else None


def cpu_baseline(args, cores):
    """The CPU oracle's training step (fp32 PyTorch-CPU restatement of the reference path + AdamW) on ONE
    synthetic scene of the same law; 1 warm-up + 2 timed steps (about 10-20 s)."""
    from oracle import star_oracle as O
    from star_amd.models import multiagent_mae as mm
    from star_amd.synthetic import make_batch
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    ref = getattr(mm, args.model)(time_stamp=args.time_stamp, mask_method=args.mask)
    c = ref._cfg()
    cfg = O.make_cfg(img=c.img, patch=c.patch, chans=c.chans, dim=c.dim, depth=c.depth, heads=c.heads, ddim=c.ddim,
                     ddepth=c.ddepth, dheads=c.dheads, T=c.T, mask=c.mask)
    sd = O.trainable(ref.state_dict())
    named = [(k, v) for k, v in sd.items()]
    opt = torch.optim.AdamW(O.weight_decay_groups(named, 0.05), lr=1e-3, betas=(0.9, 0.95))
    data = make_batch(1, args.agents, args.time_stamp, device="cpu")
    bev = data["bev_seq"].squeeze(1).permute(0, 3, 1, 2).contiguous()
    nxt = data["bev_seq_next"].permute(0, 1, 4, 2, 3).contiguous() if args.time_stamp > 1 else data["bev_seq_next"]
    times = []
    for it in range(3):
        t0 = time.perf_counter()
        loss, *_ = O.run(sd, cfg, bev, nxt, data["trans_matrices"].numpy(), data["num_agent"].numpy(), 1, mask_ratio=args.mask_ratio)
        loss.backward()
        opt.step()
        opt.zero_grad()
        times.append(time.perf_counter() - t0)
    t = sum(times[1:]) / 2
    return {"value": round(args.agents / t, 3), "unit": "BEV frames/s", "cores": cores, "kind": "port",
            "sample": "CPU oracle (fp32 PyTorch-CPU) training step incl. AdamW, 1 scene x %d agents, 1 warm-up + 2 timed steps" % args.agents}


def main():
    args = parse()
    if os.environ.get("STAR_AMD_WATCHDOG"):     # debugging aid: dump every thread's stack and exit if the run takes longer
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["STAR_AMD_WATCHDOG"]), exit=True)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local % torch.cuda.device_count())
        backend = os.environ.get("STAR_AMD_DIST_BACKEND", "nccl")   # "gloo": rehearsal of the N>1 flow where RCCL cannot run (ranks sharing one GPU)
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend=backend)
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d (WORLD_SIZE=%d)" % (args.gpus, world)
    if local >= torch.cuda.device_count():      # rehearsal on a box with fewer GPUs than ranks
        local = local % torch.cuda.device_count()
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    from star_amd import ops
    from star_amd.models import multiagent_mae as mm
    from star_amd.optim import FlatAdamW
    from star_amd.parallel import SceneParallel
    from star_amd.synthetic import make_batch
    from star_amd.utils import lr_sched
    from star_amd.utils.CoModule import CoModule

    torch.manual_seed(0)
    if args.com == "vqstar":
        from star_amd.models import VQSTAR
        args.model = "vqstar(K=8192,E=32)"
        model = VQSTAR.vqstar(time_stamp=args.time_stamp, mask_method=args.mask, num_vq_embeddings=8192, vq_embedding_dim=32)
        model = model.to(dev).set_compute_dtype(args.dtype)
        opt = FlatAdamW(model, lr=1e-3, betas=(0.9, 0.999), weight_decay=0.0, grouping="flat")
    else:
        model = getattr(mm, args.model)(time_stamp=args.time_stamp, mask_method=args.mask).to(dev).set_compute_dtype(args.dtype)
        opt = FlatAdamW(model, lr=1e-3, betas=(0.9, 0.95), weight_decay=0.05)
    model.train() if args.mode == "train" else model.eval()
    par = SceneParallel(model, opt) if world > 1 else None
    if par is not None:
        par.broadcast_parameters(0)
    cm = CoModule(model, opt, args.com, parallel=par)
    data = make_batch(args.batch, args.agents, args.time_stamp, seed=1 + rank, device=dev)   # each rank: its own scenes
    sched = types.SimpleNamespace(lr=1e-3, min_lr=0.0, warmup_epochs=4, nepoch=100)
    frames_per_step = args.batch * args.agents
    it_per_epoch = 1000
    state = {"it": 0}

    def step():
        if args.mode == "train":
            lr_sched.adjust_learning_rate(opt, state["it"] / it_per_epoch + 1, sched)
            torch.manual_seed(123 + state["it"] * world + rank)          # per-rank mask noise stream
            if args.com == "vqstar":
                loss, result, _, _ = cm.step_vqstar_completion(data, args.batch, args.mask_ratio, trainable=True)
            else:
                loss, result, _ = cm.step_mae_completion(data, args.batch, args.mask_ratio, trainable=True)
        elif args.com == "vqstar":
            loss = cm.infer_vqstar_completion(data, args.batch, args.mask_ratio)[0]
        else:
            loss, result, _, _ = cm.infer_mae_completion(data, args.batch, args.mask_ratio)
        state["it"] += 1
        return loss

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # probe step with every kernel family timed -> pick the dominant one
    ops.PROFILER = ops.Profiler()
    step()
    probe = ops.PROFILER.summary()
    ops.PROFILER = None
    dominant = max(probe, key=lambda k: probe[k]["ms"])
    for _ in range(args.warmup):
        step()
    ops.PROFILER = ops.Profiler(tags={dominant})
    sync()
    t0 = time.perf_counter()
    sample_every = 4        # HIP-event pairs around every launch of the dominant kernel on every 4th timed step
    sampled = 0
    for i in range(args.steps):
        ops.PROFILER.active = i % sample_every == 0
        sampled += int(ops.PROFILER.active)
        loss = step()
    sync()
    elapsed = time.perf_counter() - t0
    prof = ops.PROFILER.summary()[dominant]
    ops.PROFILER = None
    if world > 1:
        tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = tmax.item()
    if rank == 0:
        value = frames_per_step * world * args.steps / elapsed
        peak_f = PEAK_BF16_TFLOPS if args.dtype == "bf16" else 157.3
        # the resource that bounds the family: time at peak MFMA rate vs time at peak HBM rate for its algorithmic work
        mfma_bound = prof["flops"] > 0 and prof["flops"] / (peak_f * 1e12) >= prof["bytes"] / (PEAK_HBM_GBS * 1e9)
        if mfma_bound:
            ach = prof["flops"] / (prof["ms"] * 1e-3) / 1e12
            roof = {"kernel": dominant, "bound": "mfma", "achieved": round(ach, 2), "peak": PEAK_BF16_TFLOPS if args.dtype == "bf16" else 157.3,
                    "unit": "TFLOP/s", "frac": round(ach / (PEAK_BF16_TFLOPS if args.dtype == "bf16" else 157.3), 4), "traffic": None,
                    "launches_per_step": prof["launches"] // sampled, "avg_launch_us": round(1e3 * prof["ms"] / prof["launches"], 2),
                    "share_of_step": round(prof["ms"] / sampled / (1e3 * elapsed / args.steps), 3)}
        else:
            ach = prof["bytes"] / (prof["ms"] * 1e-3) / 1e9
            roof = {"kernel": dominant, "bound": "hbm", "achieved": round(ach, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                    "frac": round(ach / PEAK_HBM_GBS, 4), "traffic": None, "launches_per_step": prof["launches"] // sampled,
                    "avg_launch_us": round(1e3 * prof["ms"] / prof["launches"], 2),
                    "share_of_step": round(prof["ms"] / sampled / (1e3 * elapsed / args.steps), 3),
                    "mfma_tflops": round(prof["flops"] / (prof["ms"] * 1e-3) / 1e12, 1)}
        roof["timed_launches"] = prof["launches"]     # HIP-event samples inside the timed region (every 4-th step)
        roof["algorithmic_bytes_per_launch"] = round(prof["bytes"] / prof["launches"])
        roof["algorithmic_flops_per_launch"] = round(prof["flops"] / prof["launches"])
        traffic = _pmc_traffic(dominant)
        if traffic is not None:
            roof["traffic"] = traffic
        util = _pmc_mfma_util(dominant)
        if util is not None:
            roof["mfma_util_pmc"] = util
        if args.profile_all:
            tot = sum(v["ms"] for v in probe.values())
            for k, v in sorted(probe.items(), key=lambda kv: -kv[1]["ms"]):
                print("  %-14s %4d launches %8.3f ms (%4.1f%%) %8.1f TFLOP/s %8.1f GB/s" % (
                    k, v["launches"], v["ms"], 100 * v["ms"] / tot, v["flops"] / (v["ms"] * 1e-3 + 1e-12) / 1e12,
                    v["bytes"] / (v["ms"] * 1e-3 + 1e-12) / 1e9), file=sys.stderr)
        out = {
            "metric": "BEV frames/sec (5-agent scenes, %s)" % ("train" if args.mode == "train" else "infer"),
            "value": round(value, 2), "unit": "BEV frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "STAR completion %s step, %s, batch %d scenes x %d agents per GPU, 256x256x13 BEV, T=%d %s mask"
                       % (args.mode, args.model, args.batch, args.agents, args.time_stamp, args.mask),
                       "global_batch_scenes": args.batch * world, "frames_per_step": frames_per_step * world,
                       "parallelism": "dp%d (scene-sharded, RCCL grad all-reduce)" % world if world > 1 else "single GPU",
                       "loss": round(float(loss), 6)},
            "roofline": roof,
        }
        if world == 1 and not args.no_cpu_baseline and args.mode == "train" and args.com == "ind_mae":
            out["cpu_baseline"] = cpu_baseline(args, min(16, os.cpu_count() or 1))
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
