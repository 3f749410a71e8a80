import importlib
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def rt():
    """the product package (ctypes over libraytracer_amd.so)"""
    return importlib.import_module("ray-tracer_amd")


@pytest.fixture(scope="session")
def orc():
    """the CPU oracle binding (test infrastructure)"""
    from oracle import binding
    binding.build()
    return binding


@pytest.fixture(scope="session")
def models_dir(rt):
    return rt.scenes.models_dir()


@pytest.fixture(scope="session")
def golden_meta():
    with open(os.path.join(GOLDEN, "meta.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def ctx(rt):
    """one HIP context for the whole GPU session"""
    return rt.Context(0)
