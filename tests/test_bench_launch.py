"""not-gpu: `python bench.py --gpus N` (N > 1, no WORLD_SIZE) turns itself into a torch.distributed.run launch without
touching the GPU: the parent has imported neither torch nor the HIP binding when it builds the command."""
import json
import os
import subprocess
import sys

from conftest import ROOT


def _print_launch(extra, env=None):
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--print-launch"] + extra, env=e, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-800:]
    return json.loads(r.stdout.strip().splitlines()[-1])


def test_self_launch_command_and_untouched_gpu():
    d = _print_launch(["--gpus", "8", "--steps", "7", "--warmup", "3", "--dist-backend", "gloo"])
    argv = d["argv"]
    assert argv[1:3] == ["-m", "torch.distributed.run"]
    assert "--nnodes=1" in argv and "--nproc-per-node=8" in argv
    assert argv[argv.index("--master-addr") + 1] == "127.0.0.1"
    assert 1024 < int(argv[argv.index("--master-port") + 1]) < 65536
    i = argv.index(os.path.join(ROOT, "bench.py"))
    assert argv[i + 1:] == ["--gpus", "8", "--steps", "7", "--warmup", "3", "--dist-backend", "gloo"]     # the caller's flags, unchanged
    assert d["torch_imported"] is False and d["hip_imported"] is False


def test_under_torchrun_nothing_is_relaunched():
    """with WORLD_SIZE set (the driver's own torchrun form) the process is a rank: no second launcher"""
    e = {k: v for k, v in os.environ.items()}
    e.update(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--print-launch"], env=e,
                       capture_output=True, text=True, timeout=300)
    assert "torch.distributed.run" not in r.stdout
    assert r.returncode != 0 and "needs an MI355X" in (r.stderr + r.stdout)      # a rank without a GPU fails loudly
