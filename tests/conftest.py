import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _reload_switches(name="NG_"):
    """libnmrgnn_hip.so parses its NG_* path switches once; tell it when a test changes one"""
    if not name.startswith("NG_"):
        return
    from nmrgnn_amd import _lib
    if _lib._lib is not None:
        _lib._lib.ng_reload_env()


class _SwitchPatch:
    """pytest's monkeypatch with setenv / delenv that also make the engine re-read its switches"""

    def __init__(self, mp):
        self._mp = mp

    def setenv(self, name, value, *a, **kw):
        self._mp.setenv(name, value, *a, **kw)
        _reload_switches(name)

    def delenv(self, name, *a, **kw):
        self._mp.delenv(name, *a, **kw)
        _reload_switches(name)

    def __getattr__(self, item):
        return getattr(self._mp, item)


@pytest.fixture
def monkeypatch(monkeypatch):
    yield _SwitchPatch(monkeypatch)
    monkeypatch.undo()
    _reload_switches()


@pytest.fixture(scope="session")
def gpu_device():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda", 0)
