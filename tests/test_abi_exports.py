"""CPU-side checks of the drop-in boundary: the shared library loads and exports every symbol that
include/phylo_hip.h declares; without a GPU, compute entry points fail loudly (no CPU fallback)."""
import ctypes
import os
import re

import numpy as np
import pytest

from phylo_amd import _ffi

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "phylo_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(phylo_[A-Za-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported():
    lib = _ffi.load()
    names = declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), "libphylo_hip.so does not export %s" % n
    assert sorted(_ffi.EXPORTS) == names, "phylo_amd/_ffi.py EXPORTS out of sync with include/phylo_hip.h"
    assert b"gfx950" in lib.phylo_version()


def test_no_cpu_fallback_without_device():
    if _ffi.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(_ffi.PhyloError) as e:
        _ffi.Context(4, 5, 10)
    assert e.value.code == -2 and "no CPU path" in str(e.value)


def test_product_never_imports_oracle():
    """The package must not import, link or execute anything under oracle/."""
    pkg = os.path.join(ROOT, "phylo_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", ".sh")):
                src = open(os.path.join(dirpath, f), errors="ignore").read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
                assert "liboracle" not in src, f
                assert not re.search(r'#\s*include\s*[<"][^>"]*(oracle|ora_)', src), f
    # ... nor runner.py, nor anything under tools/ (profiling and probing scripts): the checker is used from tests/, from
    # __graft_entry__.smoke() and from bench.py's cpu_baseline / parity legs only
    others = [os.path.join(ROOT, "runner.py")] + [os.path.join(ROOT, "tools", f) for f in os.listdir(os.path.join(ROOT, "tools"))
                                                   if f.endswith((".py", ".sh"))]
    for path in others:
        src = open(path, errors="ignore").read()
        assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), path
        assert "liboracle" not in src, path
