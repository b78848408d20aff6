"""bench.py launched the way the driver launches it for N > 1 (python -m torch.distributed.run, one process per
rank), on the one GPU a test box has: two ranks that share GPU 0 and meet over gloo, and one rank over RCCL."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _launch(ranks, extra):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(ranks), "--steps", "3", "--warmup", "1",
           "--workload", "soup260k_bounce", "--width", "640", "--height", "360"] + extra
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=280)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout   # rank 0 prints, nobody else
    return json.loads(lines[0])


def test_two_ranks_share_the_gpu():
    line = _launch(2, ["--backend", "gloo", "--one-device"])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["steps"] == 3
    per_rank = line["config"]["rays_per_step_per_gpu"]
    # the value counts every rank's rays over the slowest rank's time
    assert 0.5 * per_rank < line["value"] * 1e6 * line["ms_per_step"] * 1e-3 / 2 < 1.5 * per_rank
    assert "cpu_baseline" not in line and line["roofline"]["frac"] <= 1.0


def test_one_rank_over_rccl():
    line = _launch(1, ["--cpu-seconds", "2", "--no-host-buffers"])
    assert line["n_gpus"] == 1 and line["value"] > 0
    assert line["cpu_baseline"]["hits_identical_to_gpu"] is True


def test_one_rank_frame_over_rccl():
    # whole frames through the C++ frame level, launched by torch.distributed.run: the RCCL id travels over the process
    # group, the framebuffer through ncclAllReduce (a world of one rank here), samples are traced in groups
    line = _launch(1, ["--frame", "--sppx", "5"])
    assert line["n_gpus"] == 1 and line["scaling"] == "strong" and line["value"] > 0
    assert line["framebuffer_exchange"]["used_rccl"] == 1
    assert line["config"]["sppx"] == 5
