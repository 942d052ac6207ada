"""bench.py's output contract on the GPU box: one JSON line with the driver's keys, the roofline block for the dominant
kernel (live HIP-event timing) and, by default at N=1, the cpu_baseline block."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*flags):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "1", *flags],
                         capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    return json.loads(lines[0])


def test_bench_line_train():
    j = _run()
    for k, v in dict(n_gpus=1, steps=4, warmup=1, higher_is_better=True, scaling="weak", dtype="bf16", data="synthetic", vs_baseline=None).items():
        assert j[k] == v, (k, j[k])
    assert j["unit"] == "BEV frames/s" and j["value"] > 100 and abs(j["value"] - 40 * 1e3 / j["ms_per_step"]) < 0.02 * j["value"]
    assert "workload" in j["config"] and "model" not in j["config"]
    r = j["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["timed_launches"] > 0 and r["avg_launch_us"] > 0 and (r["traffic"] is None or r["traffic"] > 0)
    assert r["traffic"] is None or "mask" in j["config"]["workload"]      # PMC numbers only for the workload they were measured on
    s = j["step_roofline"]      # frames/s x FLOP/frame / peak
    assert abs(s["gflop_per_frame"] - 131.77) < 0.1 and abs(s["frac_of_mfma_peak"] - j["value"] * 131.77e9 / 2.5e15) < 2e-3
    fam = {f["kernel"] for f in j["families"]}
    assert {"gemm_nt", "gemm_wgrad", "attn_bwd"} <= fam
    for f in j["families"] + j["bandwidth_path"]:
        assert 0 < f["frac"] <= 1.0, f                      # a fraction of a peak: never above it
    assert {"ce_argmax", "warp_fuse"} <= {f["kernel"] for f in j["bandwidth_path"]}
    c = j["cpu_baseline"]
    assert c["machine_physical_cores"] is None or c["machine_physical_cores"] >= 1
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and c["sample"] and c["cpu_model"]
    assert c["value"] == c["train_frames_per_s"] and c["infer_frames_per_s"] > c["train_frames_per_s"]


def test_bench_line_infer_has_no_cpu_leg():
    j = _run("--mode", "infer", "--no-cpu-baseline")
    assert "infer" in j["metric"] and "cpu_baseline" not in j and j["roofline"]["frac"] > 0
    for f in j["families"] + j["bandwidth_path"]:
        assert 0 < f["frac"] <= 1.0, f


def test_bench_line_infer_bf16x3():
    j = _run("--mode", "infer", "--no-cpu-baseline", "--dtype", "bf16x3")
    assert j["dtype"] == "bf16x3" and j["value"] > 100 and abs(j["step_roofline"]["peak_tflops"] - 2500.0 / 3) < 1.0
