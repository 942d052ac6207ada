#!/usr/bin/env python3
"""STAR scene-completion training on MI355X -- the `ind_mae` path of the reference's
completion/train_completion.py with the same flags, learning-rate schedule, log lines and checkpoint layout
(`completion_epoch_<n>.pth` holding epoch / model_state_dict / optimizer_state_dict / scheduler_state_dict /
mae_scaler_state_dict / loss).  The V2X-Sim loader of the reference needs the absent `coperception` package and
dataset class; `-d <root>` reads the same pre-processed sample files through star_amd.datasets (sparse indices, densified on
the GPU), `-d synthetic` (default) draws batches from star_amd.synthetic with the same per-batch dictionary layout.

Single GPU:   python completion/train_completion.py --com ind_mae --mae_model amortized_ind_patch8_light \\
                  --mask complement --time_stamp 1 --batch 8 --nepoch 2 --log --logpath logs
Multi GPU:    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 completion/train_completion.py ...
              (scenes are sharded over ranks; gradients are averaged over RCCL)"""
import argparse
import glob
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from star.models import VQSTAR, multiagent_mae  # noqa: E402
from star.utils import lr_sched  # noqa: E402
from star.utils.CoModule import CoModule  # noqa: E402
from star_amd.optim import FlatAdamW, add_weight_decay  # noqa: E402
from star_amd.parallel import SceneParallel  # noqa: E402
from star_amd.datasets import SparseSceneDataset, discover  # noqa: E402
from star_amd.synthetic import make_batch  # noqa: E402


class AverageMeter:
    def __init__(self, name, fmt=":f"):
        self.name, self.fmt, self.sum, self.count, self.avg = name, fmt, 0.0, 0, 0.0

    def update(self, val, n=1):
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count

    def __str__(self):
        return ("{name} {avg" + self.fmt + "}").format(name=self.name, avg=self.avg)


def check_folder(path):
    os.makedirs(path, exist_ok=True)
    return path


def main(args):
    world, rank, local = int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("STAR_AMD_DIST_BACKEND", "nccl")     # "gloo": rehearsal with ranks that share one GPU
        local = local % torch.cuda.device_count()
        torch.cuda.set_device(local)
        if backend == "nccl":
            torch.distributed.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            torch.distributed.init_process_group(backend)
    device = torch.device("cuda", local)
    torch.manual_seed(args.seed)                     # the same initial weights on every rank (rank 0's are broadcast anyway)
    if args.com not in ("ind_mae", "vqstar"):
        raise ValueError("com: %s is not supported by the MI355X path (use ind_mae or vqstar)" % args.com)
    num_agent = args.num_agent - 1 if args.no_cross_road else args.num_agent
    vq = args.com == "vqstar"
    if vq:      # reference train_completion.py:93-98,126-128: VQSTAR.vqstar + plain Adam
        model = VQSTAR.vqstar(norm_pix_loss=args.norm_pix_loss, time_stamp=args.time_stamp, mask_method=args.mask, decay=args.decay,
                              commitment_cost=args.commitment_cost, num_vq_embeddings=args.num_vq_embeddings,
                              vq_embedding_dim=args.vq_embedding_dim)
    else:
        model = multiagent_mae.__dict__[args.mae_model](norm_pix_loss=args.norm_pix_loss, time_stamp=args.time_stamp, mask_method=args.mask,
                                                        encode_partial=args.encode_partial, no_temp_emb=args.no_temp_emb,
                                                        decode_singletemp=args.decode_singletemp)
    model = model.to(device).set_compute_dtype(args.dtype)
    if args.fused_optimizer:
        if vq:
            optimizer = FlatAdamW(model, lr=args.lr, betas=(0.9, 0.999), weight_decay=0.0, grouping="flat")
        else:
            optimizer = FlatAdamW(model, lr=args.lr, betas=(0.9, 0.95), weight_decay=args.weight_decay)
        parallel = SceneParallel(model, optimizer) if world > 1 else None
        if parallel is not None:
            parallel.broadcast_parameters(0)
    else:
        assert world == 1, "multi-GPU training uses the fused optimizer's flat gradient buffer"
        optimizer = (torch.optim.Adam(model.parameters(), lr=args.lr) if vq else
                     torch.optim.AdamW(add_weight_decay(model, args.weight_decay), lr=args.lr, betas=(0.9, 0.95)))
        parallel = None
    co_module = CoModule(model, optimizer, args.com, parallel=parallel)
    torch.manual_seed(args.seed * 1000003 + rank)    # mask noise (and the EMA seeding's shuffle): every rank its own stream

    cross = "no_cross" if args.no_cross_road else "with_cross"
    model_save_path = check_folder(os.path.join(args.logpath or "logs", args.com, cross))
    start_epoch = 1
    resume = args.resume
    if args.auto_resume_path:
        files = glob.glob(os.path.join(args.auto_resume_path, args.com, cross, "completion_epoch_*.pth"))
        resume = max(files, key=os.path.getctime) if files else ""
    if resume:
        checkpoint = torch.load(resume, map_location="cpu", weights_only=True)
        start_epoch = checkpoint["epoch"] + 1
        if args.fused_optimizer:
            model.load_state_dict(checkpoint["model_state_dict"])   # parameters are views of the flat buffer: copied in place
            optimizer.load_state_dict(checkpoint["optimizer_state_dict"])
            model._weights.epoch += 1
        else:
            co_module.resume_from_cpu(checkpoint=checkpoint, device=device)
        print("Load model from {}, at epoch {}".format(resume, start_epoch - 1))
    saver = None
    if args.log and rank == 0:
        saver = open(os.path.join(model_save_path, "log_completion.txt"), "a" if resume else "w")
        saver.write("GPU number: {}\n".format(world))
        saver.write("command line: {}\n".format(" ".join(sys.argv)))
        saver.write(args.__repr__() + "\n\n")
        saver.flush()

    dataset = None
    if args.data != "synthetic":     # a pre-processed V2X-Sim split: <data>/agent<i>/<scene>_<frame>/<sample>.npy (reference layout)
        dataset = SparseSceneDataset(*discover(args.data, args.num_agent, args.no_cross_road), time_stamp=args.time_stamp,
                                     no_cross_road=args.no_cross_road)
        print("Training dataset size:", len(dataset))
    iters = args.iters_per_epoch if dataset is None else dataset.steps_per_epoch(args.batch, world)

    def batches(epoch):
        if dataset is None:
            for it in range(iters):
                yield make_batch(args.batch, num_agent, args.time_stamp, seed=(epoch * iters + it) * world + rank + 1, device=device), args.batch
        else:
            yield from dataset.loader(args.batch, shuffle=True, num_workers=args.nworker, device=device, seed=epoch, rank=rank, world=world)

    for epoch in range(start_epoch, args.nepoch + 1):
        lr = co_module.optimizer.param_groups[0]["lr"]
        if rank == 0:
            print("Epoch {}, learning rate {}".format(epoch, lr))
        if saver:
            saver.write("epoch: {}, lr: {}\t".format(epoch, lr))
        running = AverageMeter("Total loss", ":.6f")
        co_module.model.train()
        for it, (data, bsz) in enumerate(batches(epoch)):
            lr_sched.adjust_learning_rate(co_module.optimizer, it / iters + epoch, args)
            if vq:
                loss, reconstruction, _, perplexity = co_module.step_vqstar_completion(data, bsz, args.mask_ratio, trainable=True)
            else:
                loss, reconstruction, _ = co_module.step_mae_completion(data, bsz, args.mask_ratio, trainable=True)
            running.update(loss)
            if it % args.logstep == 0 and rank == 0:
                print("**Epoch {} -- iterstep {}** : imme loss {:.6f}, running loss {:.6f}, current lr {:.6f}".format(
                    epoch, it, loss, running.avg, co_module.optimizer.param_groups[0]['lr']))
                if vq:
                    print("imme perplexity {:.6f}".format(float(perplexity)))
        if saver:
            saver.write("{}\n".format(running))
            saver.flush()
            torch.save({"epoch": epoch, "model_state_dict": co_module.model.state_dict(),
                        "optimizer_state_dict": co_module.optimizer.state_dict(),
                        "scheduler_state_dict": None,   # the reference calls None.state_dict() here and crashes
                        "mae_scaler_state_dict": co_module.mae_loss_scaler.state_dict(), "loss": running.avg},
                       os.path.join(model_save_path, "completion_epoch_" + str(epoch) + ".pth"))
    if saver:
        saver.close()
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    p = argparse.ArgumentParser()
    p.add_argument("-d", "--data", default="synthetic", type=str,
                   help="'synthetic', or the root of a pre-processed V2X-Sim split (agent<i>/<scene>_<frame>/ sample files, reference format)")
    p.add_argument("--batch", default=4, type=int, help="Batch size (scenes per GPU)")
    p.add_argument("--nepoch", default=100, type=int)
    p.add_argument("--nworker", default=2, type=int)
    p.add_argument("--lr", default=0.001, type=float)
    p.add_argument("--log", action="store_true")
    p.add_argument("--logpath", default="")
    p.add_argument("--resume", default="", type=str)
    p.add_argument("--com", default="ind_mae", type=str)
    p.add_argument("--no_cross_road", action="store_true")
    p.add_argument("--num_agent", default=6, type=int)
    p.add_argument("--mask_ratio", default=0.50, type=float)
    p.add_argument("--mask", default="random", type=str, choices=["random", "complement"])
    p.add_argument("--warmup_epochs", default=4, type=int)
    p.add_argument("--min_lr", default=0., type=float)
    p.add_argument("--time_stamp", default=2, type=int)
    p.add_argument("--mae_model", default="amortized_ind_patch8_light", type=str)
    p.add_argument("--norm_pix_loss", default=False, type=bool)
    p.add_argument("--weight_decay", default=0.05, type=float)
    p.add_argument("--wandb", action="store_true", help="accepted for CLI compatibility; ignored")
    p.add_argument("--logstep", default=200, type=int)
    p.add_argument("--auto_resume_path", default="", type=str)
    p.add_argument("--encode_partial", action="store_true")
    p.add_argument("--no_temp_emb", action="store_true")
    p.add_argument("--decode_singletemp", action="store_true")
    p.add_argument("--decay", default=0., type=float, help="used in vector quantization")
    p.add_argument("--commitment_cost", default=0.25, type=float, help="Used in vector quantization")
    p.add_argument("--num_vq_embeddings", default=512, type=int, help="Used in vector quantization")
    p.add_argument("--vq_embedding_dim", default=64, type=int, help="Used in vector quantization")
    # extras of the MI355X build
    p.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    p.add_argument("--iters_per_epoch", default=100, type=int, help="synthetic batches per epoch")
    p.add_argument("--no_fused_optimizer", dest="fused_optimizer", action="store_false")
    p.add_argument("--seed", default=0, type=int)
    a = p.parse_args()
    print(a)
    main(a)
