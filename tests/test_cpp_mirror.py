"""The C++ host mirror (include/bio.hpp) over the C ABI, exercised by a plain g++ program
(tests/cpp/bio_test.cpp, built by `make`) that reads like the reference's Example tests."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "cpp", "bio_test")


def _run(*args):
    assert os.path.exists(BIN), "tests/cpp/bio_test is missing: run `make`"
    return subprocess.run([BIN, *args], capture_output=True, text=True, timeout=120)


def test_cpp_mirror_host_logic():
    import torch

    extra = ["--cpu", "--expect-no-device"] if not torch.cuda.is_available() else ["--cpu"]
    r = _run(*extra)
    assert r.returncode == 0, r.stdout + r.stderr


@pytest.mark.gpu
def test_cpp_mirror_golden_vectors_on_gpu():
    r = _run("--gpu")
    assert r.returncode == 0, r.stdout + r.stderr
