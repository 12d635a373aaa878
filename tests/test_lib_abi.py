"""CPU: the C-ABI library loads and exports every symbol include/cmlpl.h declares (no compute calls)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "cmlpl.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(cmlpl_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported():
    from cmlpl_amd import _lib, build_ext
    if build_ext.needs_build():
        build_ext.build(verbose=False)
    lib = _lib.load()
    names = _declared()
    assert set(names) == set(_lib.EXPORTS), (names, _lib.EXPORTS)
    for n in names:
        assert hasattr(lib, n), n
    assert lib.cmlpl_abi_version() == _lib.ABI_VERSION


def test_layout_matches_reference_parameter_counts():
    from cmlpl_amd import _lib
    L = _lib.layout(_lib.Shape(60, 20, 20, 103, 9))
    assert sum(L.param_numel) == 552329
    assert sum(L.param_numel[:10]) == 207881                 # live parameters
    assert L.cls_in == 2624                                  # tools/models.py:127
    assert all(o % 4 == 0 for o in L.param_off)
    L2 = _lib.layout(_lib.Shape(103, 11, 11, 103, 9))
    assert L2.cls_in == 64 * 2 * 2 + 1024
    assert _lib.load().cmlpl_layout(C.byref(_lib.Shape(60, 3, 3, 103, 9)), C.byref(_lib.Layout())) == -2
    assert _lib.load().cmlpl_layout(C.byref(_lib.Shape(60, 20, 20, 103, 65)), C.byref(_lib.Layout())) == -2


def test_missing_library_fails_loudly(tmp_path):
    from cmlpl_amd import _lib
    saved = _lib._lib
    _lib._lib = None
    try:
        with pytest.raises(_lib.CmlplLibraryError):
            _lib.load(str(tmp_path / "nope.so"))
    finally:
        _lib._lib = saved


def test_engine_refuses_cpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from cmlpl_amd import TrainEngine, NetShape
    with pytest.raises(RuntimeError):
        TrainEngine(NetShape(), 32, 32)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "cmlpl_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                txt = open(os.path.join(dp, f)).read()
                assert "oracle" not in txt.replace("no CPU fallback", ""), os.path.join(dp, f)
