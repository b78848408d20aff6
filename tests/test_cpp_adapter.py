"""The C++ host adapter (include/rtgi_hip_tracer.hpp): a small C++ program drives the tracer through the
renderer's class interfaces and checks it against the oracle. Built by __graft_entry__.build()."""
import os
import subprocess

import pytest

BIN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cpp", "adapter_test")


def _run(mode):
    assert os.path.exists(BIN), "run __graft_entry__.build() (or make -C tests/cpp) first"
    r = subprocess.run([BIN, mode], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "adapter_test: OK" in r.stdout
    return r


def test_adapter_host_side():
    r = _run("host")
    assert "requires exactly one positive integral value" in r.stderr   # the malformed 'bvh sah' is reported


@pytest.mark.gpu
def test_adapter_on_gpu():
    r = _run("gpu")
    assert "hits" in r.stdout


@pytest.mark.gpu
def test_cpp_wavefront_frame_example(tmp_path):
    # examples/wavefront_frame.cpp: the whole camera -> trace -> shade -> framebuffer loop from C++ on the C ABI
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "wavefront_frame")
    assert os.path.exists(exe), "run __graft_entry__.build() (or make -C examples) first"
    lines = []
    for k in range(2):
        out = str(tmp_path / ("frame%d.ppm" % k))
        r = subprocess.run([exe, "320", "180", "3", out], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stdout + r.stderr
        assert "framebuffer identical to the stage-by-stage one" in r.stdout  # the same frame through hip_bvh_frame_tracer
        lines.append(r.stdout.strip().splitlines()[0])
        data = open(out, "rb").read()
        assert data.startswith(b"P6\n320 180\n255\n") and len(data) == len(b"P6\n320 180\n255\n") + 320 * 180 * 3
    lit = float(lines[0].split("lit bytes ")[1].split("%")[0])
    assert lit > 60.0  # a closed, lit room: most of the image is neither black nor missed
    assert lines[0].split("checksum ")[1].split(";")[0] == lines[1].split("checksum ")[1].split(";")[0]
