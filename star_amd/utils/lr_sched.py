"""Per-iteration learning-rate schedule: linear warm-up, then half-cycle cosine.
Mirrors star/utils/lr_sched.py:9-21 (same name, arguments, `lr_scale` handling)."""
import math


def adjust_learning_rate(optimizer, epoch, args):
    if epoch < args.warmup_epochs:
        lr = args.lr * epoch / args.warmup_epochs
    else:
        span = args.nepoch - args.warmup_epochs
        # evaluation order (pi * elapsed, then / span) is kept as in the reference: the table must match bit for bit
        lr = args.min_lr + (args.lr - args.min_lr) * 0.5 * (1.0 + math.cos(math.pi * (epoch - args.warmup_epochs) / span))
    for group in optimizer.param_groups:
        group["lr"] = lr * group["lr_scale"] if "lr_scale" in group else lr
    return lr
