"""Synthetic V2X-Sim-shaped batches with the tuple layout the reference's training loop assembles
(completion/train_completion.py:225-275): per-batch dict with
  bev_seq [A*B,1,H,W,C] float32 (agent-major rows a*B+b), bev_seq_next [A*B,T-1,H,W,C] (or [A*B,0]),
  bev_seq_teacher [A*B,1,H,W,C], trans_matrices [B,A,A,4,4] float64 on the HOST, num_agent [B,A] int64.
Law: 3 % occupancy, teacher 8 %, planar rigid poses with yaw U(-pi,pi) and
x,y U(-20,20) m, identity on the diagonal."""
import math

import numpy as np
import torch


def rigid_poses(batch, agents, seed):
    rng = np.random.RandomState(seed)
    t = np.zeros((batch, agents, agents, 4, 4), dtype=np.float64)
    for b in range(batch):
        for i in range(agents):
            for j in range(agents):
                if i == j:
                    t[b, i, j] = np.eye(4)
                    continue
                yaw, tx, ty = rng.uniform(-math.pi, math.pi), rng.uniform(-20, 20), rng.uniform(-20, 20)
                c, s = math.cos(yaw), math.sin(yaw)
                t[b, i, j] = [[c, -s, 0, tx], [s, c, 0, ty], [0, 0, 1, 0], [0, 0, 0, 1]]
    return torch.from_numpy(t)


def make_batch(batch, agents=5, time_stamp=1, size=256, chans=13, seed=1, occupancy=0.03, device="cuda", num_agent=None):
    g = torch.Generator(device="cpu").manual_seed(seed)
    ab = agents * batch
    bev = (torch.rand(ab, 1, size, size, chans, generator=g) < occupancy).float()
    if time_stamp > 1:
        nxt = (torch.rand(ab, time_stamp - 1, size, size, chans, generator=g) < occupancy).float()
    else:
        nxt = torch.zeros(ab, 0)
    teacher = torch.maximum(bev, (torch.rand(ab, 1, size, size, chans, generator=g) < 0.08).float())
    na = torch.full((batch, agents), agents, dtype=torch.int64) if num_agent is None else torch.as_tensor(num_agent, dtype=torch.int64).reshape(batch, 1).repeat(1, agents)
    return {
        "bev_seq": bev.to(device), "bev_seq_next": nxt.to(device), "bev_seq_teacher": teacher.to(device),
        "trans_matrices": rigid_poses(batch, agents, seed), "num_agent": na.to(device),
    }


def structured_bev(n, size=256, chans=13, seed=1, walls=(6, 14), boxes=(8, 20), noise=0.002):
    """Learnable occupancy maps for the trained-weight parity fixtures (random bits cannot be compressed, so a model trained on
    `make_batch` only learns "free everywhere"): per map a few straight wall segments with a contiguous height span, some small
    filled boxes (vehicles) two to four channels tall, and a little salt noise.  -> float32 [n, chans, size, size] in {0,1},
    about 4-8 % occupied.  Pure numpy RandomState: the same bits in the fixture generator and on the GPU box."""
    rng = np.random.RandomState(seed)
    out = np.zeros((n, chans, size, size), dtype=np.float32)
    sc = size / 64.0
    for m in range(n):
        for _ in range(rng.randint(walls[0], walls[1] + 1)):
            x0, y0 = rng.uniform(0, size, 2)
            ang = rng.uniform(0, math.pi)
            ln = rng.uniform(8, 40) * sc
            z0 = rng.randint(0, chans - 3)
            z1 = min(chans, z0 + rng.randint(3, 8))
            t = np.arange(0, ln, 0.5)
            xs = np.clip(np.round(x0 + t * math.cos(ang)).astype(np.int64), 0, size - 1)
            ys = np.clip(np.round(y0 + t * math.sin(ang)).astype(np.int64), 0, size - 1)
            out[m, z0:z1, ys[:, None], xs[:, None]] = 1.0
        for _ in range(rng.randint(boxes[0], boxes[1] + 1)):
            w, h = max(2, int(round(rng.uniform(2, 5) * sc))), max(2, int(round(rng.uniform(2, 5) * sc)))
            x0, y0 = rng.randint(0, size - w), rng.randint(0, size - h)
            z0 = rng.randint(0, chans - 4)
            out[m, z0:z0 + rng.randint(2, 5), y0:y0 + h, x0:x0 + w] = 1.0
        out[m] = np.maximum(out[m], (rng.rand(chans, size, size) < noise).astype(np.float32))
    return torch.from_numpy(out)


def make_structured_batch(batch, agents=5, time_stamp=1, size=256, chans=13, seed=1, device="cuda", num_agent=None):
    """`make_batch` with `structured_bev` maps: same dict layout, channels-last frames."""
    ab = agents * batch
    frames = structured_bev(ab * (time_stamp + 1), size, chans, seed)
    bev = frames[:ab].permute(0, 2, 3, 1).unsqueeze(1).contiguous()
    if time_stamp > 1:
        nxt = frames[ab:ab * time_stamp].reshape(ab, time_stamp - 1, chans, size, size).permute(0, 1, 3, 4, 2).contiguous()
    else:
        nxt = torch.zeros(ab, 0)
    teacher = torch.maximum(bev, frames[ab * time_stamp:].permute(0, 2, 3, 1).unsqueeze(1).contiguous())
    na = torch.full((batch, agents), agents, dtype=torch.int64) if num_agent is None else torch.as_tensor(num_agent, dtype=torch.int64).reshape(batch, 1).repeat(1, agents)
    return {"bev_seq": bev.to(device), "bev_seq_next": nxt.to(device), "bev_seq_teacher": teacher.to(device),
            "trans_matrices": rigid_poses(batch, agents, seed), "num_agent": na.to(device)}
