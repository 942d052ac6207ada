#!/usr/bin/env python3
"""STAR scene-completion evaluation on MI355X -- the `ind_mae` and `vqstar` paths of the reference's
completion/test_completion.py: load `checkpoint["model_state_dict"]`, run `infer_mae_completion` (ego slot = the
agent's own input), accumulate the three-scale occupancy IoU per batch and print the average.
Synthetic batches stand in for V2X-Sim (see train_completion.py)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from star.models import VQSTAR, multiagent_mae  # noqa: E402
from star.utils.CoModule import CoModule  # noqa: E402
from star.utils.metrics import Metrics  # noqa: E402
from star_amd.datasets import SparseSceneDataset, discover  # noqa: E402
from star_amd.synthetic import make_batch  # noqa: E402


def main(args):
    world, rank, local = int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0))
    if world > 1:
        # replicas only: rank r evaluates batches r, r + world, ...; the per-batch IoUs are gathered to every rank and averaged
        # in batch order, so the result is the single-process one bit for bit (the reference averages PER-BATCH IoUs,
        # star/utils/metrics.py:210-218, so summing confusion matrices over ranks would give a different number)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("STAR_AMD_DIST_BACKEND", "nccl")     # "gloo": rehearsal with ranks that share one GPU
        local = local % torch.cuda.device_count()
        torch.cuda.set_device(local)
        if backend == "nccl":
            torch.distributed.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            torch.distributed.init_process_group(backend)
    device = torch.device("cuda", local)
    torch.manual_seed(args.seed)       # without --resume the weights are random: at least the same ones in every process
    num_agent = args.num_agent - 1 if args.no_cross_road else args.num_agent
    if args.com == "vqstar":    # reference test_completion.py:108-113
        model = VQSTAR.vqstar(norm_pix_loss=args.norm_pix_loss, time_stamp=args.time_stamp, mask_method=args.mask, decay=args.decay,
                              commitment_cost=args.commitment_cost, num_vq_embeddings=args.num_vq_embeddings,
                              vq_embedding_dim=args.vq_embedding_dim)
    else:
        model = multiagent_mae.__dict__[args.mae_model](norm_pix_loss=args.norm_pix_loss, time_stamp=args.time_stamp, mask_method=args.mask)
    if args.resume:
        checkpoint = torch.load(args.resume, map_location="cpu", weights_only=True)
        model.load_state_dict(checkpoint["model_state_dict"])
        print("Load model from {}, at epoch {}".format(args.resume, checkpoint["epoch"]))
    model = model.to(device).set_compute_dtype(args.dtype)
    model.eval()
    optimizer = torch.optim.AdamW(model.parameters(), lr=args.lr)
    co_module = CoModule(model, optimizer, args.com)
    if args.data != "synthetic":     # a pre-processed V2X-Sim split in the reference's on-disk format, in file order
        dataset = SparseSceneDataset(*discover(args.data, args.num_agent, args.no_cross_road), time_stamp=args.time_stamp,
                                     no_cross_road=args.no_cross_road)
        print("Validation dataset size:", len(dataset))
        args.iters = dataset.steps_per_epoch(args.batch)
        batches = dataset.loader(args.batch, shuffle=False, num_workers=args.nworker, device=device)
    else:
        batches = ((it, make_batch(args.batch, num_agent, args.time_stamp, seed=10_000 + it, device=device), args.batch)
                   for it in range(args.iters) if it % world == rank)
    evaluator = Metrics(2, args.iters)
    losses = []
    if args.data != "synthetic":
        batches = ((it,) + tuple(b) for it, b in enumerate(batches) if it % world == rank)
    for it, data, bsz in batches:
        torch.manual_seed(args.seed + it)        # the mask noise of batch `it` does not depend on which rank evaluates it
        if args.com == "vqstar":
            loss, reconstruction, ind_pred, perplexity, encodings = co_module.infer_vqstar_completion(data, bsz, args.mask_ratio)
        else:
            loss, reconstruction, ind_pred, latent = co_module.infer_mae_completion(data, bsz, args.mask_ratio)
        teacher = data["bev_seq_teacher"].squeeze(1).permute(0, 3, 1, 2)
        evaluator.add_batch(reconstruction, teacher.contiguous())
        evaluator.update_IoU()
        losses.append(loss)
    if world > 1:
        mine = [(rank + i * world, iou, l) for i, (iou, l) in enumerate(zip(evaluator.every_batch_IoU, losses))]
        parts = [None] * world
        torch.distributed.all_gather_object(parts, mine)
        rows = sorted((r for p in parts for r in p), key=lambda r: r[0])
        evaluator.every_batch_IoU = [r[1] for r in rows]
        losses = [r[2] for r in rows]
    iou = evaluator.get_average_IoU()
    if rank == 0:
        if args.verbose:
            for i, (v, l) in enumerate(zip(evaluator.every_batch_IoU, losses)):
                print("batch {}: loss {:.9f} IoU {:.9f} {:.9f} {:.9f}".format(i, l, *v))
        print("Average loss {:.6f}".format(sum(losses) / len(losses)))
        print("Occupancy IoU  scale 1: {:.9f}  scale 1/2: {:.9f}  scale 1/4: {:.9f}".format(*iou))
    if world > 1:
        torch.distributed.destroy_process_group()
    return iou


if __name__ == "__main__":
    p = argparse.ArgumentParser()
    p.add_argument("-d", "--data", default="synthetic", type=str)
    p.add_argument("--batch", default=1, type=int)
    p.add_argument("--lr", default=0.001, type=float)
    p.add_argument("--resume", default="", type=str)
    p.add_argument("--com", default="ind_mae", type=str)
    p.add_argument("--no_cross_road", action="store_true")
    p.add_argument("--num_agent", default=6, type=int)
    p.add_argument("--mask_ratio", default=0.50, type=float)
    p.add_argument("--mask", default="random", type=str, choices=["random", "complement"])
    p.add_argument("--time_stamp", default=2, type=int)
    p.add_argument("--mae_model", default="amortized_ind_patch8_light", type=str)
    p.add_argument("--norm_pix_loss", default=False, type=bool)
    p.add_argument("--decay", default=0., type=float)
    p.add_argument("--commitment_cost", default=0.25, type=float)
    p.add_argument("--num_vq_embeddings", default=512, type=int)
    p.add_argument("--vq_embedding_dim", default=64, type=int)
    p.add_argument("--dtype", default="bf16", choices=["bf16", "fp32", "bf16x3"],
                   help="bf16x3: bf16 hi/lo pairs, three MFMA products per multiply -- the float32 path's decisions and IoU at 2.9x its speed")
    p.add_argument("--iters", default=10, type=int, help="synthetic batches to evaluate")
    p.add_argument("--nworker", default=2, type=int)
    p.add_argument("--verbose", action="store_true", help="print every batch's loss and IoU")
    p.add_argument("--seed", default=0, type=int, help="mask-noise seed (random masking draws noise at inference too)")
    main(p.parse_args())
