"""The C-ABI library builds, loads, and exports every symbol include/nmrgnn_hip.h declares
(no compute calls: this runs without a GPU)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "nmrgnn_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ng_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as ge
    from nmrgnn_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        ge.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    syms = declared_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/nmrgnn_hip.h but not exported"
    assert set(_lib.SIGNATURES) == set(syms), set(_lib.SIGNATURES) ^ set(syms)
    assert lib.ng_abi_version() == 9


def test_engine_refuses_to_run_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from nmrgnn_amd import _lib
    from nmrgnn_amd.engine import Engine
    from nmrgnn_amd.hypers import HyperParameters, declare_gnn_space
    with pytest.raises(_lib.NGError):
        Engine(declare_gnn_space(HyperParameters()), 10)


def test_product_package_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "nmrgnn_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in re.sub(r'""".*?"""', "", src, flags=re.S).replace("# oracle", ""), f
