"""gpu: the N > 1 path of bench.py executed for real on a one-GPU box -- two ranks started by bench.py's own self-launch
(torch.distributed.run, rendezvous on 127.0.0.1), both on GPU 0 (CLHIP_BENCH_ALL_ON_GPU0=1), gloo for the barrier and the
max-over-ranks reduction (RCCL refuses two ranks on one device; the data path has no collective, SURVEY.md section 8e: the
independent unit is one Soapy device per channel, soapy_api/SoapyCariboulite.cpp:46-69).  Exactly what the driver's N = 2, 4, 8
runs execute except for the backend name."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT


def _run(extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["CLHIP_BENCH_ALL_ON_GPU0"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--steps", "3",
                        "--warmup", "1", "--settle", "1", "--no-cpu"] + extra, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, f"rank 0 prints ONE line, got {len(lines)}: {r.stdout[-800:]}"
    return json.loads(lines[0])


@pytest.mark.gpu
def test_two_ranks_headline_workload():
    d = _run(["--log2-samples", "20"])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["scaling"] == "weak" and d["unit"] == "Msamples/s"
    assert d["config"]["samples_per_gpu_per_step"] == 1 << 20
    # whole-job value = both ranks' samples over the max-over-ranks time
    assert abs(d["value"] - 2 * (1 << 20) * 3 / (d["ms_per_step"] * 3e-3) / 1e6) / d["value"] < 1e-2
    assert "cpu_baseline" not in d                       # rank 0 at N = 1 only
    assert d["roofline"]["bound"] == "hbm" and d["roofline"]["achieved"] > 0


@pytest.mark.gpu
def test_two_ranks_sharded_config4():
    d = _run(["--workload", "c4", "--streams", "6", "--log2-samples", "17"])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["streams_per_gpu"] == 3


@pytest.mark.gpu
def test_two_ranks_sharded_config4_with_the_fan_out_step():
    """--fanout: rank 0 holds every stream's raw buffer and hands each rank its own before the timed region (the one real exchange step,
    SURVEY.md section 8e).  On a node that is clfan_scatter_streams over RCCL; in this one-GPU rehearsal the same schedule goes through
    torch.distributed's point-to-point calls (gloo, the messages through host copies) -- bench.py's own code around it, the barriers and
    the scatter's separate clock, is what runs here for real."""
    d = _run(["--workload", "c4", "--streams", "6", "--log2-samples", "17", "--fanout"])
    assert d["n_gpus"] == 2 and d["config"]["streams_per_gpu"] == 3
    assert d["config"]["fanout_scatter_s"] is not None and 0 < d["config"]["fanout_scatter_s"] < 60


@pytest.mark.gpu
def test_two_ranks_at_the_soapy_boundary():
    """--pcie: each rank owns a stream group of its own (8 Soapy devices here), feeds it host bytes and reads host samples; the
    line's value is both ranks' samples over the max-over-ranks time."""
    d = _run(["--pcie", "--pcie-streams", "8"])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["roofline"]["bound"] == "pcie"
    assert d["config"]["streams_per_gpu"] == 8 and d["config"]["group"]["errors"] == 0 and d["config"]["group"]["single_reads"] == 0
    assert abs(d["value"] - 2 * 8 * 131072 / (d["ms_per_step"] * 1e-3) / 1e6) / d["value"] < 1e-2


@pytest.mark.gpu
def test_rccl_calls_of_the_n_gt_1_path_at_world_size_one():
    """The driver's N = 2, 4, 8 runs use the nccl (= RCCL) backend: init_process_group with a device id, barrier, the max-reduction on
    a DEVICE tensor, destroy.  Two ranks cannot share one GPU under RCCL, so those calls run here at world size 1 under
    torch.distributed.run (--force-dist), everything else of the line as at N = 1."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
                        "--master-port", "29517", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-dist", "--dist-backend", "nccl",
                        "--steps", "3", "--warmup", "1", "--settle", "1", "--no-cpu", "--log2-samples", "20"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["value"] > 0
