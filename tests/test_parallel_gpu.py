"""SURVEY 8(e) on the one-GPU box: two ranks that SHARE the GPU, over gloo (RCCL refuses two ranks on one device), run as
child processes under torch.distributed.run.  What is checked is the data-parallel CONTRACT, not the transport:
  * 2 ranks x B/2 scenes through SceneParallel == 1 rank x B scenes (gradients, loss, parameters after the step);
  * the backward-overlapped hand-over keeps the replicas identical over three steps;
  * multi-rank evaluation (per-batch IoUs gathered in batch order) prints the single-process IoU, digit for digit
    (reference star/utils/metrics.py:210-218 averages per-batch IoUs).
N >= 2 over RCCL runs on the driver's 8-GPU node; the RCCL call path with one rank is in test_model_gpu.py."""
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _torchrun(script, *args, port, nproc=2, timeout=170):
    env = dict(os.environ, PYTHONPATH=ROOT, STAR_AMD_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, script)] + list(args)
    return subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)


def test_two_ranks_half_batches_equal_one_rank_whole_batch():
    out = _torchrun("tools/rehearse_dp.py", "equiv", port=29611)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    assert "rank 0 equiv OK" in out.stdout and "rank 1 equiv OK" in out.stdout, out.stdout[-1500:]


def test_overlapped_bucket_handover_keeps_replicas_identical():
    out = _torchrun("tools/rehearse_dp.py", "async", port=29612)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    assert out.stdout.count("params equal to rank 0: True") == 2, out.stdout[-1500:]


def test_multi_rank_evaluation_reproduces_single_process_iou():
    args = ["--com", "ind_mae", "--batch", "1", "--num_agent", "3", "--time_stamp", "2", "--mask", "random", "--iters", "4", "--dtype", "fp32", "--verbose"]
    env = dict(os.environ, PYTHONPATH=ROOT)
    one = subprocess.run([sys.executable, os.path.join(ROOT, "completion", "test_completion.py")] + args, env=env, capture_output=True,
                         text=True, timeout=170)
    assert one.returncode == 0, one.stderr[-3000:]
    two = _torchrun("completion/test_completion.py", *args, port=29613)
    assert two.returncode == 0, (two.stdout[-1500:], two.stderr[-3000:])
    pat = re.compile(r"Occupancy IoU .*")
    a, b = pat.findall(one.stdout), pat.findall(two.stdout)
    assert len(a) == 1 and len(b) == 1 and a == b, (one.stdout[-800:], two.stdout[-800:])
    la, lb = re.findall(r"Average loss .*", one.stdout), re.findall(r"Average loss .*", two.stdout)
    assert la == lb and len(la) == 1
