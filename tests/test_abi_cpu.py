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
    assert lib.sgam_abi_version() == _lib.ABI_VERSION == 10
    assert b"gfx950" in lib.sgam_build_info()
    # the build stamp (csrc/build_info.hip): commit of the library's sources + digest over sources and flags
    commit, digest = lib.sgam_build_commit().decode(), lib.sgam_build_digest().decode()
    assert commit and commit != "unknown" and len(digest) == 12
    assert commit.encode() in lib.sgam_build_info() and digest.encode() in lib.sgam_build_info()


def test_every_declared_symbol_is_exported_and_bound():
    lib = ctypes.CDLL(_lib.LIB_PATH)
    declared = _declared_symbols()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/sgam_hip.h but not exported"
    assert sorted(_lib.PROTOTYPES.keys()) == declared, "ctypes prototypes and the header disagree"


def test_every_declared_symbol_is_documented_for_the_reference_side():
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    assert [s for s in _declared_symbols() if s not in doc] == []


def test_conv_desc_layout_matches_header():
    text = open(os.path.join(ROOT, "include", "sgam_hip.h")).read()
    body = re.search(r"typedef struct sgam_conv_desc \{(.*?)\} sgam_conv_desc;", text, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = [f.strip().lstrip("*") for decl in re.findall(r"int32_t ([^;]+);", body) for f in decl.split(",")]
    assert fields == [n for n, _ in _lib.ConvDesc._fields_]
    assert ctypes.sizeof(_lib.ConvDesc) == 4 * len(_lib.ConvDesc._fields_) == 88


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
    # fused attention: C = 256 and n a multiple of 256 only; workspace = K / V^T fragments + 8 key ranges of partial O
    assert lib.sgam_attention_f32x_workspace_bytes(4096, 256) == 2 * 4096 * 256 * 4 + 8 * 4096 * 256 * 4 + 8 * 4096 * 8
    assert lib.sgam_attention_f32x_workspace_bytes(4096, 512) == -1
    assert lib.sgam_attention_f32x_workspace_bytes(4000, 256) == -1
    assert lib.sgam_attention_f32x(None, None, None, 768, 4096, 256, 0.0625, None, 256, None, 0, None) == -1
    assert lib.sgam_attention_h16_workspace_bytes(4096, 256) == 2 * 4096 * 256 * 2 + 8 * 4096 * 256 * 4 + 8 * 4096 * 8
    assert lib.sgam_attention_h16_workspace_bytes(4096, 128) == -1
    assert lib.sgam_attention_h16(None, None, None, 1, 768, 4096, 256, 0.0625, None, 256, None, 0, None) == -1
    # batched forms: fewer key ranges per image as the batch fills the chip by itself (four or more 64 x 64 images: two ranges of 2048 keys)
    assert lib.sgam_attention_f32x_batched_workspace_bytes(4096, 256, 1) == lib.sgam_attention_f32x_workspace_bytes(4096, 256)
    assert lib.sgam_attention_f32x_batched_workspace_bytes(4096, 256, 8) == 8 * (2 * 4096 * 256 * 4 + 2 * 4096 * 256 * 4 + 2 * 4096 * 8)
    assert lib.sgam_attention_f32x_batched_workspace_bytes(4096, 256, 4) == 4 * (2 * 4096 * 256 * 4 + 2 * 4096 * 256 * 4 + 2 * 4096 * 8)
    assert lib.sgam_attention_f32x_batched_workspace_bytes(16384, 256, 4) == 4 * lib.sgam_attention_f32x_workspace_bytes(16384, 256)
    assert lib.sgam_attention_f32x_workspace_bytes(16384, 256) == 2 * 16384 * 256 * 4 + 8 * 16384 * 256 * 4 + 8 * 16384 * 8   # <= 2048 keys / range
    assert lib.sgam_attention_h16_batched_workspace_bytes(4096, 256, 2) == 2 * (2 * 4096 * 256 * 2 + 4 * 4096 * 256 * 4 + 4 * 4096 * 8)
    assert lib.sgam_attention_f32x_batched_workspace_bytes(4096, 256, 0) == -1
    assert lib.sgam_attention_f32x_batched(None, None, None, 768, 4096, 256, 4, 0.0625, None, 256, None, 0, None) == -1
    assert lib.sgam_attention_h16_batched(None, None, None, 1, 768, 4096, 256, 4, 0.0625, None, 256, None, 0, None) == -1


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.SgamHipError, match="no CPU fallback"):
        _lib.load()


def _desc(**kw):
    base = dict(B=1, Hi=64, Wi=64, Cin=128, Ho=64, Wo=64, N=128, KH=3, KW=3, stride=1, pad_t=1, pad_l=1, upsample2x=0,
                lda=128, ldb=1152, ldc=128, ldr=0, n_valid=128, bias_per_row=0)
    base.update(kw)
    return _lib.ConvDesc(**base)


def test_split_fp32_plan_queries_without_gpu():
    """the host-side planning logic of the split-fp32 family (no launch): which kernel a descriptor gets, where the
    GroupNorm statistics of its output come from, how split-K ranges are aligned"""
    lib = _lib.load()
    ref = ctypes.byref
    d = _desc(plan_bm=64, plan_bn=128, plan_ksplit=1)
    assert lib.sgam_conv2d_f32x_uses_halo(ref(d)) == 1 and lib.sgam_conv2d_f32x_gn_fusable(ref(d)) == 1
    assert lib.sgam_conv2d_f32x_stats_chunks(ref(d)) > 0 and lib.sgam_conv2d_f32x_stats_mode(ref(d)) == 1
    # nearest-2x upsampling conv: halo kernel, but no GroupNorm precedes it
    u = _desc(Hi=32, Wi=32, upsample2x=1, plan_bm=64, plan_bn=128, plan_ksplit=1)
    assert lib.sgam_conv2d_f32x_uses_halo(ref(u)) == 1 and lib.sgam_conv2d_f32x_gn_fusable(ref(u)) == 0
    # stride 2 / 1x1 / maps that do not tile into 8x8 patches: generic kernel
    for g in (_desc(Ho=32, Wo=32, stride=2, pad_t=0, pad_l=0, plan_bm=64, plan_bn=128, plan_ksplit=1),
              _desc(KH=1, KW=1, pad_t=0, pad_l=0, ldb=128, plan_bm=64, plan_bn=128, plan_ksplit=1),
              _desc(Hi=20, Wi=20, Ho=20, Wo=20, plan_bm=64, plan_bn=128, plan_ksplit=1)):
        assert lib.sgam_conv2d_f32x_uses_halo(ref(g)) == 0
    # the heuristic's (64, 64) tile on a 3x3 / s1 / p1 shape is the halo kernel's 64-channel tile (2 x 2 wavefronts), not the generic kernel
    assert lib.sgam_conv2d_f32x_uses_halo(ref(_desc())) == 1
    # halo plans split K on whole channel slabs (9 taps): 4 slabs of 9 taps -> ranges of 9, 18 or 36
    bm, bn, ks = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
    s = _desc(plan_bm=64, plan_bn=128, plan_ksplit=3)
    assert lib.sgam_conv2d_f32x_plan(ref(s), ref(bm), ref(bn), ref(ks)) == 0 and ks.value == 2      # 18 + 18
    assert lib.sgam_conv2d_f32x_workspace_bytes(ref(s)) == 2 * 64 * 64 * 128 * 4
    assert lib.sgam_conv2d_f32x_stats_chunks(ref(s)) == 64 * 64 * 128 // 1024                     # from the combine
    # rejected descriptors: layout constraints of the fragment-ordered B operand and the 16-byte epilogue
    assert lib.sgam_conv2d_f32x_workspace_bytes(ref(_desc(ldb=1150))) == -1
    assert lib.sgam_conv2d_f32x_workspace_bytes(ref(_desc(Cin=100, lda=100, ldb=928))) == -1      # taps need whole slabs
    assert lib.sgam_conv2d_f32x_workspace_bytes(ref(_desc(n_valid=126))) == -1
    assert lib.sgam_conv2d_f32x_plan(ref(_desc(plan_bm=96, plan_bn=128)), ref(bm), ref(bn), ref(ks)) < 0
    # entry points refuse NULL operands before any launch
    assert lib.sgam_conv2d_nhwc_f32x(ref(d), None, 1.0, None, 1.0, None, None, None, None, 0, None) == -1
    assert lib.sgam_conv2d_gn_nhwc_f32x(ref(d), None, None, None, None, 1, None, 1.0, None, None, None, None, None, 0, None) == -1


def test_group_major_combine_needs_power_of_two_groups():
    """the group-major split-K combine folds a group over cpg / 4 lanes with an xor butterfly, so it is only
    selected when channels-per-group divides the tile and is a power of two.  N = 384 / 768 (cpg = 12 / 24, a `ch_mult` with
    a 3) on a 16 x 16 map under split-K must fall back to the row-major combine's chunk count or to 0 (two-pass statistics)."""
    lib = _lib.load()
    ref = ctypes.byref
    for n in (384, 768):
        d = _desc(Hi=16, Wi=16, Ho=16, Wo=16, Cin=n, N=n, lda=n, ldb=9 * n, ldc=n, n_valid=n, plan_bm=64, plan_bn=128, plan_ksplit=4)
        hw = 16 * 16
        row_major = hw * n // 1024 if (1024 % n == 0 and (hw * n) % 1024 == 0) else 0
        assert lib.sgam_conv2d_f32x_stats_chunks(ref(d)) == row_major == 0
        assert lib.sgam_conv2d_h16_stats_chunks(ref(d)) in (0, row_major)
    # the shipped shapes keep the group-major form: N = 512 (cpg 16) on 16 x 16 -> TC = 32 or 16, at most 16 chunks
    d = _desc(Hi=16, Wi=16, Ho=16, Wo=16, Cin=512, N=512, lda=512, ldb=9 * 512, ldc=512, n_valid=512, plan_bm=64, plan_bn=128, plan_ksplit=4)
    assert 0 < lib.sgam_conv2d_f32x_stats_chunks(ref(d)) <= 16


def test_tsdf_argument_validation_without_gpu():
    lib = _lib.load()
    g = _lib.TsdfGrid(0.01, 0.03, (ctypes.c_int32 * 3)(0, 0, 0), (ctypes.c_int32 * 3)(4, 4, 4))
    bad = _lib.TsdfGrid(0.0, 0.03, (ctypes.c_int32 * 3)(0, 0, 0), (ctypes.c_int32 * 3)(4, 4, 4))
    srcs = (_lib.TsdfSrc * 2)()
    args = (8, 8, 10.0, 10.0, 4.0, 4.0, 20.0, 1, None, None, None, None, 16, None, None, 16, None, None, None)
    assert lib.sgam_tsdf_integrate_srcs_f32(ctypes.byref(g), srcs, 2, *args) == -1          # no state pointers
    assert lib.sgam_tsdf_integrate_srcs_f32(ctypes.byref(bad), srcs, 2, *args) == -1
    assert lib.sgam_tsdf_integrate_srcs_f32(ctypes.byref(g), srcs, 9, *args) == -1          # more than 8 sources
    assert ctypes.sizeof(_lib.TsdfSrc) == 16 + 128
    assert lib.sgam_tsdf_ray_mult_f32(8, 8, 10.0, 10.0, 4.0, 4.0, None, None) == -1
    assert lib.sgam_tsdf_raycast_depth_f32(ctypes.byref(g), 8, 8, 10.0, 10.0, 4.0, 4.0, None, 0.1, 4.0, None, None, None,
                                           None, None, None) == -1
