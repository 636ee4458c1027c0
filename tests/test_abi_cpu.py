"""CPU-side checks of the C-ABI library: it loads without a GPU and exports every symbol that
include/tt.h declares; the Python binding table mirrors the header.  No compute calls here."""
import ctypes
import re

import pytest

from conftest import ROOT


def _declared():
    text = (ROOT / "include" / "tt.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tt_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def libtt():
    from twotowermlretrieval_amd import build, _lib
    build.build()
    return _lib.lib()


def test_header_declares_functions():
    names = _declared()
    assert "tt_score_topk_f32" in names and "tt_last_error" in names and len(names) >= 6


def test_library_exports_every_declared_symbol(libtt):
    for name in _declared():
        assert hasattr(libtt, name), f"libtt.so does not export {name}"


def test_binding_table_matches_header():
    from twotowermlretrieval_amd import _lib
    assert sorted(_lib.SIGNATURES) == _declared()


def test_version_and_error_strings(libtt):
    assert b"gfx950" in libtt.tt_version()
    assert isinstance(libtt.tt_last_error(), bytes)


def test_workspace_query_needs_no_gpu(libtt):
    n = libtt.tt_score_topk_workspace_bytes(32, 1_000_000, 256, 10)
    assert n > 0 and n % 8 == 0
    assert libtt.tt_score_topk_workspace_bytes(0, 10, 256, 10) == 0


def test_argument_validation_without_gpu(libtt):
    from twotowermlretrieval_amd import _lib
    rc = libtt.tt_score_topk_f32(None, 4, 100, None, 10, 5, 0, ctypes.c_void_p(16), ctypes.c_void_p(16), None, 0, None)
    assert rc == _lib.TT_ERR_UNSUPPORTED and b"d=100" in libtt.tt_last_error()
    rc = libtt.tt_score_topk_f32(None, 4, 256, None, 10, 65, 0, ctypes.c_void_p(16), ctypes.c_void_p(16), None, 0, None)
    assert rc == _lib.TT_ERR_UNSUPPORTED
    rc = libtt.tt_topk_merge(None, None, -1, 0, 5, None, None, None)
    assert rc == _lib.TT_ERR_BAD_SHAPE


def test_product_refuses_cpu_tensors():
    import torch
    import twotowermlretrieval_amd as tt
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        tt.score_topk(torch.zeros(2, 256), torch.zeros(10, 256), 5)


def test_product_never_imports_oracle():
    pkg = ROOT / "twotowermlretrieval_amd"
    for f in list(pkg.rglob("*.py")) + list(pkg.rglob("*.hip")) + list(pkg.rglob("*.h")):
        src = f.read_text()
        assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
        assert "tt_oracle" not in src or f.suffix != ".py", f
