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
PEAK_F32_TFLOPS = 157.3        # fp32-input MFMA = the fp32 vector rate (same guide)
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
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32", "bf16x3"],
                    help="bf16x3 (inference only): bf16 hi/lo pairs, three bf16 MFMA products per multiply -- the float32 path's decisions")
    ap.add_argument("--mode", default="train", choices=["train", "infer"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile-all", action="store_true", help="print the per-kernel-family time table to stderr")
    return ap.parse_args()


# kernel family (ops.Profiler tag) -> kernel-name prefixes in the rocprofv3 PMC summary committed under profiles/
_FAMILY_KERNELS = {"gemm_nt": ("void gemm_nt_fast_kernel", "void gemm_nt_wreg_kernel", "void gemm_nt_kernel"), "gemm_wgrad": ("void gemm_tn_", "wgrad_reduce_kernel"), "gemm_res_ln": ("void gemm_res_ln_kernel",),
                   "attn_fwd": ("void attn_fwd_",), "attn_bwd": ("void attn_bwd_", "void attn_delta_"),
                   "layernorm_fwd": ("void ln_fwd_",), "layernorm_bwd": ("void ln_bwd_", "ln_bwd_reduce"),
                   "ce_argmax": ("void ce_argmax_",), "warp_fuse": ("warp_fuse_", "wf_pack_bits"), "patchify": ("void patchify_kernel",)}


def _pmc_file(name, workload):
    """A committed PMC summary (tools/pmc_traffic.sh / tools/pmc_mfma.sh over this same bench command), or None when it is
    absent or was collected for a DIFFERENT workload string: counters cannot be collected inside the timed run, so they are
    read from profiles/ -- but only for the workload they were measured on."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", name)), reverse=True):     # newest round first
        with open(path) as f:
            d = json.load(f)
        if d.get("workload") == workload:
            return d
    return None


def _pmc_traffic(family, workload):
    """HBM-side bytes per launch of the family's kernels (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, gfx950
    FETCH_SIZE correction applied)."""
    d = _pmc_file("r*_pmc_traffic.json", workload)
    if d is None or family not in _FAMILY_KERNELS:
        return None
    tot, n = 0.0, 0
    for name, v in d["kernels"].items():
        if name.startswith(_FAMILY_KERNELS[family]):
            tot += (v["fetch_bytes_per_launch"] + v["write_bytes_per_launch"]) * v["launches"]
            n += v["launches"] if not name.startswith(("wgrad_reduce", "ln_bwd_reduce", "void attn_delta_", "wf_pack_bits")) else 0
    return round(tot / n) if n else None


def _pmc_mfma_util(family, workload):
    """Fraction of SIMD cycles with the MFMA pipe busy for the family (one more --pmc pass)."""
    d = _pmc_file("r*_pmc_mfma.json", workload)
    if d is None:
        return None
    v = d["families"].get(family)
    return round(v["mfma_util_busy_cycles"], 4) if v and v.get("mfma_util_busy_cycles") is not None else None


def flops_per_frame(c, mask, mask_ratio, train, joint=False):
    """algorithmic FLOPs (2 x MACs) per output frame of the configured model; train = 3 x forward."""
    import math
    L, D, Dd, V, T = c.L, c.dim, c.ddim, c.V, c.T
    keep = int(L * (1 - mask_ratio)) if mask == "random" else math.ceil(L / T)

    def block(dm, l):
        return 2 * l * 12 * dm * dm + 4 * l * l * dm

    l_enc = keep if c.encode_partial else L
    if joint:
        # encoder on every token, decoder blocks, then two conv heads: stage s = bilinear x2 + 3x3 conv at (grid 2^s)^2 pixels
        # (ConvPred 512 -> 256 -> 128 -> C at patch 8: mae_base.py:16-54)
        enc = 2 * L * V * D + c.depth * block(D, L) + 2 * keep * D * c.inter
        dec = T * (2 * keep * c.inter * D + 2 * keep * D * Dd) + c.ddepth * block(Dd, L)
        px, cin = c.grid * c.grid, Dd
        stages = int(math.log2(c.patch))
        head = 0
        for st in range(stages):
            px *= 4
            cout = c.chans if st == stages - 1 else cin // 2
            head += 2 * px * 9 * cin * cout
            cin = cout
        return (T * enc + dec + 2 * head) * (3 if train else 1)
    enc = 2 * l_enc * V * D + c.depth * block(D, l_enc) + 2 * keep * D * c.inter
    dec = T * (2 * keep * c.inter * D + 2 * keep * D * Dd) + c.ddepth * block(Dd, L) + 2 * L * Dd * 2 * V
    return (T * enc + dec) * (3 if train else 1)


def _cpu_info():
    """(model string, physical cores of the whole machine) from lscpu, /proc/cpuinfo as the fallback (report both)."""
    import subprocess
    model, cores = "unknown", None
    try:
        out = subprocess.run(["lscpu"], capture_output=True, text=True, timeout=10).stdout
        f = {k.strip(): v.strip() for k, v in (line.split(":", 1) for line in out.splitlines() if ":" in line)}
        model = f.get("Model name", model)
        if "Core(s) per socket" in f and "Socket(s)" in f:
            cores = int(f["Core(s) per socket"]) * int(f["Socket(s)"])
    except Exception:
        pass
    if model == "unknown" or cores is None:
        try:
            seen = set()
            phys = core = None
            with open("/proc/cpuinfo") as fh:
                for line in fh:
                    if line.startswith("model name") and model == "unknown":
                        model = line.split(":", 1)[1].strip()
                    elif line.startswith("physical id"):
                        phys = line.split(":", 1)[1].strip()
                    elif line.startswith("core id"):
                        core = line.split(":", 1)[1].strip()
                        seen.add((phys, core))
            if cores is None and seen:
                cores = len(seen)
        except OSError:
            pass
    return model, cores


def cpu_baseline(args, cores, budget_s=100.0):
    """the CPU oracle (fp32 PyTorch-CPU restatement of the reference path, pinned to the reference's goldens)
    on this box's host cores, same synthetic law: 3 warm-up + 10 timed iterations, median -- the training step (forward +
    backward + AdamW over timm's groups) and the C1 inference leg (1 scene x 5 agents), each on ONE scene.  `cores` = the
    threads used: the box's CPU share for one GPU (16; os.cpu_count() shows the whole machine).  Bounded: a leg stops
    timing early when it has used its half of `budget_s` (the count actually timed is reported)."""
    import statistics

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
    opt = torch.optim.AdamW(O.weight_decay_groups(list(sd.items()), 0.05), lr=1e-3, betas=(0.9, 0.95))
    data = make_batch(1, args.agents, args.time_stamp, device="cpu")
    bev = data["bev_seq"].squeeze(1).permute(0, 3, 1, 2).contiguous()
    nxt = data["bev_seq_next"].permute(0, 1, 4, 2, 3).contiguous() if args.time_stamp > 1 else data["bev_seq_next"]
    tr, na = data["trans_matrices"].numpy(), data["num_agent"].numpy()

    def train_step():
        loss = O.run(sd, cfg, bev, nxt, tr, na, 1, mask_ratio=args.mask_ratio)[0]
        loss.backward()
        opt.step()
        opt.zero_grad()

    def infer_step():
        with torch.no_grad():
            O.run(sd, cfg, bev, nxt, tr, na, 1, mask_ratio=args.mask_ratio, infer=True)

    def timed(fn):
        t_leg = time.perf_counter()
        times = []
        for it in range(13):
            t0 = time.perf_counter()
            fn()
            if it >= 3:
                times.append(time.perf_counter() - t0)
            if time.perf_counter() - t_leg > budget_s / 2 and len(times) >= 3:
                break
        return statistics.median(times), len(times)

    t_train, n_train = timed(train_step)
    t_inf, n_inf = timed(infer_step)
    leg = "train" if args.mode == "train" else "infer"
    cpu_model, phys = _cpu_info()
    return {"value": round(args.agents / (t_train if leg == "train" else t_inf), 3), "unit": "BEV frames/s", "cores": cores, "kind": "port",
            "cpu_model": cpu_model, "machine_physical_cores": phys, "machine_logical_cpus": os.cpu_count(),
            "cores_note": "cores = threads used = the box's CPU share for one GPU (its OMP_NUM_THREADS); the machine's own core count is beside it",
            "train_frames_per_s": round(args.agents / t_train, 3), "infer_frames_per_s": round(args.agents / t_inf, 3),
            "sample": "CPU oracle (fp32 PyTorch-CPU, %d threads), 1 scene x %d agents: training step incl. AdamW (median of %d after 3 "
                      "warm-up) and inference = config C1 (median of %d after 3 warm-up); value = the %s leg"
                      % (cores, args.agents, n_train, n_inf, leg)}


def main():
    args = parse()
    if os.environ.get("STAR_AMD_WATCHDOG"):     # debugging aid: dump every thread's stack and exit if the run takes longer
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["STAR_AMD_WATCHDOG"]), exit=True)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    import torch.distributed as dist
    if world > 1 or os.environ.get("STAR_AMD_FORCE_DP") == "1":
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29544")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
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
    elif args.model.startswith("amortized_joint"):
        # the joint (latent-level fusion) variant, SURVEY 8(f) rank 1: torch AdamW over timm's decay groups, as train_completion.py builds it
        from star_amd.optim import add_weight_decay
        args.com = "joint_mae"
        model = getattr(mm, args.model)(time_stamp=args.time_stamp, mask_method=args.mask).to(dev).set_compute_dtype(args.dtype)
        opt = torch.optim.AdamW(add_weight_decay(model, 0.05), lr=1e-3, betas=(0.9, 0.95))
    else:
        model = getattr(mm, args.model)(time_stamp=args.time_stamp, mask_method=args.mask).to(dev).set_compute_dtype(args.dtype)
        opt = FlatAdamW(model, lr=1e-3, betas=(0.9, 0.95), weight_decay=0.05)
    model.train() if args.mode == "train" else model.eval()
    # STAR_AMD_FORCE_DP=1 (measurement): the data-parallel hooks and collectives with ONE rank (launch under torch.distributed.run)
    force_dp = os.environ.get("STAR_AMD_FORCE_DP") == "1" and dist.is_initialized()
    par = SceneParallel(model, opt, force=force_dp) if (world > 1 or force_dp) else None
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
        elif args.com == "joint_mae":       # the reference's joint model has no inference(): its forward without the autograd edge
            with torch.no_grad():
                loss = model(*cm._unpack(data), args.batch, mask_ratio=args.mask_ratio)[0]
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

    # one plain step (first launches load code objects), then a probe step with every kernel family timed -> the dominant one
    step()
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
        # bf16x3: one multiply is three bf16 MFMA products, so the peak for its ALGORITHMIC flops is a third of the bf16 pipe's
        peak_f = {"bf16": PEAK_BF16_TFLOPS, "bf16x3": round(PEAK_BF16_TFLOPS / 3, 1)}.get(args.dtype, PEAK_F32_TFLOPS)

        def roofline_of(tag, v, samples=None):
            """Roofline entry of one kernel family from its HIP-event times: the bound is whichever of flops/peak and
            bytes/peak is the longer time for its ALGORITHMIC work (every operand and result once)."""
            mfma_bound = v["flops"] > 0 and v["flops"] / (peak_f * 1e12) >= v["bytes"] / (PEAK_HBM_GBS * 1e9)
            tf = v["flops"] / (v["ms"] * 1e-3) / 1e12
            gb = v["bytes"] / (v["ms"] * 1e-3) / 1e9
            r = {"kernel": tag, "bound": "mfma" if mfma_bound else "hbm", "achieved": round(tf, 2) if mfma_bound else round(gb, 1),
                 "peak": peak_f if mfma_bound else PEAK_HBM_GBS, "unit": "TFLOP/s" if mfma_bound else "GB/s",
                 "frac": round(tf / peak_f, 4) if mfma_bound else round(gb / PEAK_HBM_GBS, 4), "traffic": None,
                 "avg_launch_us": round(1e3 * v["ms"] / v["launches"], 2), "mfma_tflops": round(tf, 1), "hbm_gbs": round(gb, 1),
                 "algorithmic_bytes_per_launch": round(v["bytes"] / v["launches"]), "algorithmic_flops_per_launch": round(v["flops"] / v["launches"])}
            if samples:
                r["launches_per_step"] = v["launches"] // samples
                r["timed_launches"] = v["launches"]      # HIP-event samples inside the timed region (every 4-th step)
                r["share_of_step"] = round(v["ms"] / samples / (1e3 * elapsed / args.steps), 3)
            return r

        workload = "STAR completion %s step, %s, batch %d scenes x %d agents per GPU, 256x256x13 BEV, T=%d, %s mask%s" % (
            args.mode, args.model, args.batch, args.agents, args.time_stamp, args.mask,
            " ratio %.2f" % args.mask_ratio if args.mask == "random" else " (ratio 1-1/T: mask_ratio is ignored, reference Q1)")
        roof = roofline_of(dominant, prof, sampled)
        roof["traffic"] = _pmc_traffic(dominant, workload)            # null unless measured for THIS workload
        util = _pmc_mfma_util(dominant, workload)
        if util is not None:
            roof["mfma_util_pmc"] = util
        # every family of the probe step (one step, event pair per launch, side stream included), largest first
        families = [roofline_of(k, v) for k, v in sorted(probe.items(), key=lambda kv: -kv[1]["ms"]) if v["ms"] > 0]
        for f_, (k, v) in zip(families, sorted(probe.items(), key=lambda kv: -kv[1]["ms"])):
            f_["launches_per_step"], f_["ms_per_step"] = v["launches"], round(v["ms"], 3)
        if args.profile_all:
            tot = sum(v["ms"] for v in probe.values())
            for k, v in sorted(probe.items(), key=lambda kv: -kv[1]["ms"]):
                print("  %-14s %4d launches %8.3f ms (%4.1f%%) %8.1f TFLOP/s %8.1f GB/s" % (
                    k, v["launches"], v["ms"], 100 * v["ms"] / tot, v["flops"] / (v["ms"] * 1e-3 + 1e-12) / 1e12,
                    v["bytes"] / (v["ms"] * 1e-3 + 1e-12) / 1e9), file=sys.stderr)
        fpf = flops_per_frame(model._cfg(), args.mask, args.mask_ratio, args.mode == "train", joint=args.com == "joint_mae")
        out = {
            "metric": "BEV frames/sec (5-agent scenes, %s)" % ("train" if args.mode == "train" else "infer"),
            "value": round(value, 2), "unit": "BEV frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": workload,
                       "global_batch_scenes": args.batch * world, "frames_per_step": frames_per_step * world,
                       "parallelism": "dp%d (scene-sharded, RCCL grad all-reduce)" % world if world > 1 else "single GPU",
                       "loss": round(float(loss), 6)},
            "roofline": roof,
            # the whole step against the MFMA peak = frames/s x algorithmic FLOP per frame / (N x peak)
            "step_roofline": {"gflop_per_frame": round(fpf / 1e9, 2), "achieved_tflops": round(value * fpf / 1e12 / world, 1),
                              "peak_tflops": peak_f, "frac_of_mfma_peak": round(value * fpf / 1e12 / world / peak_f, 4)},
            "families": families[:6],
            # north star: achieved HBM GB/s of the gather / warp / fusion path -- its families whatever their share of the step
            # (bytes = what each launch moves in its own operand formats, star_amd/ops.py; HIP events of the probe step)
            "bandwidth_path": [f_ for f_ in families if f_["kernel"] in ("ce_argmax", "warp_fuse", "patchify", "layernorm_fwd", "layernorm_bwd")],
        }
        for f_ in out["bandwidth_path"]:
            f_["traffic"] = _pmc_traffic(f_["kernel"], workload)
        if world == 1 and not args.no_cpu_baseline and args.com == "ind_mae":
            out["cpu_baseline"] = cpu_baseline(args, min(16, os.cpu_count() or 1))
        print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
