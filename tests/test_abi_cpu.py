"""CPU: the C-ABI library loads without a GPU and exports every symbol include/sgam_hip.h declares; the ctypes
prototypes cover exactly that set; argument validation (no compute) returns the documented error codes."""
import ctypes
import os
import re

import pytest

from sgam_neurips22_amd import _lib

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "sgam_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sgam_[a-z0-9_]+)\s*\(", text)))


def test_library_is_built_and_loads():
    assert os.path.exists(_lib.LIB_PATH), "run `python -m sgam_neurips22_amd.build`"
    lib = _lib.load()
    assert lib.sgam_abi_version() == 2
    assert b"gfx950" in lib.sgam_build_info()


def test_every_declared_symbol_is_exported_and_bound():
    lib = ctypes.CDLL(_lib.LIB_PATH)
    declared = _declared_symbols()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/sgam_hip.h but not exported"
    assert sorted(_lib.PROTOTYPES.keys()) == declared, "ctypes prototypes and the header disagree"


def test_conv_desc_layout_matches_header():
    text = open(os.path.join(ROOT, "include", "sgam_hip.h")).read()
    body = re.search(r"typedef struct sgam_conv_desc \{(.*?)\} sgam_conv_desc;", text, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = [f.strip() for decl in re.findall(r"int32_t ([^;]+);", body) for f in decl.split(",")]
    assert fields == [n for n, _ in _lib.ConvDesc._fields_]


def test_argument_validation_without_gpu():
    lib = _lib.load()
    d = _lib.ConvDesc(B=1, Hi=8, Wi=8, Cin=30, Ho=8, Wo=8, N=64, KH=3, KW=3, stride=1, pad_t=1, pad_l=1, upsample2x=0,
                      lda=32, ldb=288, ldc=64, ldr=0, n_valid=64, bias_per_row=0)
    assert lib.sgam_conv2d_workspace_bytes(ctypes.byref(d)) == -1       # Cin % 4 != 0
    d.Cin = 32
    assert lib.sgam_conv2d_workspace_bytes(ctypes.byref(d)) >= 0
    bm, bn, ks = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
    assert lib.sgam_conv2d_plan(ctypes.byref(d), ctypes.byref(bm), ctypes.byref(bn), ctypes.byref(ks)) == 0
    assert (bm.value, bn.value) in ((128, 128), (64, 128), (64, 64)) and ks.value >= 1
    assert lib.sgam_groupnorm_workspace_bytes(1, 64, 100) == -1         # C % 128 != 0
    assert lib.sgam_groupnorm_workspace_bytes(1, 64, 128) > 0
    assert lib.sgam_vq_workspace_bytes(16, 256, 4096) >= 0
    assert lib.sgam_vq_workspace_bytes(16, 250, 4096) == -1
    assert lib.sgam_softmax_rows_f32(None, 4, 4, 4, 1.0, None) == -1    # NULL pointer -> SGAM_EINVAL, no launch


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.SgamHipError, match="no CPU fallback"):
        _lib.load()
