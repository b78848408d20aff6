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
