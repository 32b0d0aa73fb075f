"""pytest configuration: registers the `gpu` marker and puts the repo root on sys.path."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


import pytest


@pytest.fixture(autouse=True)
def _reach_the_single_pass_kernels(monkeypatch):
    """libkeyes_hip sends small groups of large images down the banded path (one workgroup per image cannot fill
    the GPU); the parity tests hash a few images at a time and are about the single-pass kernels, so they lift that
    threshold.  tests/test_gpu_parity.py::test_small_groups_choose_a_path_and_all_paths_agree covers the default."""
    monkeypatch.setenv("KE_FUSED_MIN_IMAGES", "1")
