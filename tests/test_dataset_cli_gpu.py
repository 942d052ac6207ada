"""The training and evaluation CLIs on a pre-processed split in the reference's on-disk format (sparse sample files under
agent<i>/<scene>_<frame>/): discovery, sparse loading, on-device densification, training steps, checkpoint, evaluation."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_train_and_test_completion_on_sparse_split(tmp_path):
    from star_amd.datasets import voxel as V
    rng = np.random.RandomState(0)
    split = tmp_path / "train"
    n_agents = 6
    for a in range(n_agents):
        for sc in range(2):
            for fr in range(3):
                d = split / ("agent%d" % a) / ("%d_%d" % (sc, fr))
                d.mkdir(parents=True)
                if a == 5 and sc == 1:
                    V.write_sample(str(d / "0.npy"), None, None, None, 0)        # an empty agent slot
                    continue
                n = 4000
                vox = np.stack([rng.randint(0, 256, n), rng.randint(0, 256, n), rng.randint(0, 13, n)], 1)
                tea = np.concatenate([vox, np.stack([rng.randint(0, 256, n), rng.randint(0, 256, n), rng.randint(0, 13, n)], 1)])
                trans = np.tile(np.eye(4), (n_agents, 1, 1))
                trans[:, 0, 3] = rng.uniform(-20, 20, n_agents)
                V.write_sample(str(d / "0.npy"), vox, tea, trans, 6 if sc == 0 else 5, a)
    logs = tmp_path / "logs"
    env = dict(os.environ, PYTHONPATH=ROOT)
    cmd = [sys.executable, os.path.join(ROOT, "completion", "train_completion.py"), "-d", str(split), "--com", "ind_mae", "--batch", "2",
           "--nepoch", "1", "--time_stamp", "2", "--mask", "random", "--nworker", "0", "--log", "--logpath", str(logs), "--logstep", "1"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=280)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "Training dataset size: 6" in out.stdout and out.stdout.count("imme loss") == 3
    ckpt = logs / "ind_mae" / "with_cross" / "completion_epoch_1.pth"
    assert ckpt.exists()
    sd = torch.load(str(ckpt), map_location="cpu", weights_only=True)
    assert sd["epoch"] == 1 and np.isfinite(sd["loss"])
    cmd = [sys.executable, os.path.join(ROOT, "completion", "test_completion.py"), "-d", str(split), "--com", "ind_mae", "--batch", "2",
           "--time_stamp", "2", "--mask", "random", "--nworker", "0", "--resume", str(ckpt)]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=280)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "Validation dataset size: 6" in out.stdout and "Occupancy IoU" in out.stdout
