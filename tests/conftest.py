import importlib.util
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "gpu-lossless-compression_amd")
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "gpu_long: opt-in sweeps on the GPU box (minutes): run with -m gpu_long; never part of "
                                       "-m gpu or -m 'not gpu'")


def pytest_collection_modifyitems(config, items):
    """gpu_long tests run only when asked for by name (-m gpu_long): they are neither CPU tests nor part of the driver's
    -m gpu run"""
    if "gpu_long" in (config.getoption("-m") or ""):
        return
    keep, drop = [], []
    for it in items:
        (drop if it.get_closest_marker("gpu_long") else keep).append(it)
    if drop:
        config.hook.pytest_deselected(items=drop)
        items[:] = keep


def load_pkg_module(name):
    """The package directory has a hyphen in its name (fixed by the task), so
    its modules are loaded by path."""
    if name in sys.modules:
        return sys.modules[name]
    spec = importlib.util.spec_from_file_location(name, os.path.join(PKG, name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(scope="session")
def glc():
    return load_pkg_module("glc_binding")


@pytest.fixture(scope="session")
def cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU is visible (tests marked gpu must run on the MI355X box)")
    return torch.device("cuda:0")
