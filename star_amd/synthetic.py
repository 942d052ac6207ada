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
