"""Tensor-level wrappers over the C ABI of libsgam_hip.so.

torch is used here only as plumbing: device allocations (`torch.empty`), the current HIP stream and
`data_ptr()`.  Every arithmetic result comes from a hand-written HIP kernel; a non-CUDA tensor raises
(there is no CPU path in the product).

Internal activation layout is NHWC fp32: tensors of shape (B, H, W, C), contiguous.
"""
import ctypes
import os

import torch

from . import _lib
from ._lib import ConvDesc, SgamHipError, check

from ._opscore import *        # noqa: F401,F403,E402
from ._opscore import _c, _dense_nhwc, _f32c, _need_cuda, _p, _stream  # noqa: F401,E402


# How fp32 convolutions / GEMMs are evaluated:
#   "split" : fp16 matrix cores on exact hi+lo splits of both operands, fp32 accumulate (conv_f32x.hip) — fp32-class
#             accuracy at 3/16 of the fp32-MFMA issue time; the default parity path
#   "mfma"  : fp32-in MFMA (conv_gemm.hip) — bit-for-bit an fp32 fmaf chain; the reference implementation of the
#             parity path and the one the fused GroupNorm prologue / autotuned plans were built on
F32_MODE = os.environ.get("SGAM_F32_MODE", "split")
# the other 1x1 convs / GEMMs on the whole-K-panel kernel of csrc/gemm_gn_f32x.hip where they fit.  Rounds 2 - 4: 0.7 % slower in the
# frame than the generic kernel under its tuned plans (it lost proj_out by 3.5 us), so opt-in; round 5: proj_out is fused into the
# attention's merge, what is left are the shapes the panel wins (nin_shortcut, quant / post_quant convs): 373.2 -> 375.0 frames/s
# (A / B x 3, scripts/r05u.sh), full-model parity unchanged: default.  SGAM_PANEL_GEMM=0: the generic kernel
PANEL_GEMM = os.environ.get("SGAM_PANEL_GEMM", "1") == "1"
PANEL_MIN_WGS = int(os.environ.get("SGAM_PANEL_MIN_WGS", "1"))    # ... for shapes of at least this many 64 x 128 tiles
# split-mode convolutions also emit the GroupNorm statistics of their output from the epilogue (no statistics pass)
FUSE_GN_STATS = os.environ.get("SGAM_FUSE_GN_STATS", "1") == "1"
# ... and normalise(+swish) their INPUT while staging it (halo-staged 3x3 kernel): no stand-alone normalise pass
FUSE_GN_APPLY = os.environ.get("SGAM_FUSE_GN_APPLY", "1") == "1"


_RANGE_FLAGS = {}          # (device index) -> int32[1] flag tensor; entries are never freed (captured graphs hold their address)
_RANGE_FLAG_ACTIVE = None  # index of the device whose flag is registered with the library right now


def _dev_index(device):
    d = torch.device(device)
    if d.type != "cuda":
        raise SgamHipError(f"range_flag: {d} is not a HIP device")
    return torch.cuda.current_device() if d.index is None else d.index


def range_flag(device):
    """the split path's range flag of `device` (one int32 per GPU), registered with the library.  'cuda' and 'cuda:<current>'
    name the same flag; a flag is never reallocated or freed, because kernels captured into HIP graphs keep writing to its
    address.  The library holds ONE registered pointer (one process drives one GPU — distributed.py); registering another
    device's flag is allowed, but graphs captured under the previous registration keep reporting to the previous flag, so
    callers that switch devices drop their graphs (VQModel does, through `range_flag_epoch`)."""
    global _RANGE_FLAG_ACTIVE, RANGE_FLAG_EPOCH
    i = _dev_index(device)
    t = _RANGE_FLAGS.get(i)
    if t is None:
        t = _RANGE_FLAGS[i] = torch.zeros((1,), device=torch.device("cuda", i), dtype=torch.int32)
    if _RANGE_FLAG_ACTIVE != i:
        check(_lib.load().sgam_f32x_set_range_flag(_p(t)), "sgam_f32x_set_range_flag")
        _RANGE_FLAG_ACTIVE = i
        RANGE_FLAG_EPOCH += 1
    return t


RANGE_FLAG_EPOCH = 0       # bumped whenever the registered pointer changes: graphs captured before are stale


def f32x_range_tripped(reset=True, device=None):
    """True when a split-fp32 kernel wrote a non-finite output since the last reset (synchronises the device).  Reads the
    flag of `device` (default: the one registered with the library)."""
    i = _RANGE_FLAG_ACTIVE if device is None else _dev_index(device)
    t = _RANGE_FLAGS.get(i)
    if t is None:
        return False
    hit = bool(t.item())
    if hit and reset:
        t.zero_()
    return hit


def set_f32_mode(mode):
    global F32_MODE
    if mode not in ("split", "mfma"):
        raise ValueError(mode)
    F32_MODE = mode


class SplitWeight:
    """fp32 matrix (N, K) pre-split into fp16 hi / lo halves of scale * w in MFMA-fragment order
    [Np/32][Kp/32][256 pieces][8 halfs] (N, K rounded up to 32; piece = ((plane*2 + k-step)*2 + k-half)*32 + row): the
    B operand of one v_mfma_f32_32x32x16_f16 is one contiguous kilobyte (conv_f32x.hip)."""
    __slots__ = ("planes", "scale", "shape")

    def __init__(self, planes, scale, n=None, k=None, transient=False):
        np_, kp = planes.shape[0] * 32, planes.shape[1] * 32
        self.planes, self.scale, self.shape = planes, scale, (np_ if n is None else n, kp if k is None else k)

    def stride(self, dim):
        return self.planes.shape[1] * 32 if dim == 0 else 1       # logical row length (ldb), in K elements

    @property
    def dtype(self):
        return "f32x"

    @property
    def is_cuda(self):
        return self.planes.is_cuda


class FragWeightH16:
    """16-bit conv weight in MFMA-fragment order [Np/32][K/32][128 pieces][8 halfs] (csrc/h16_halo.hip)"""
    __slots__ = ("planes", "shape")

    def __init__(self, planes):
        self.planes, self.shape = planes, (planes.shape[0] * 32, planes.shape[1] * 32)

    def stride(self, dim):
        return self.shape[1] if dim == 0 else 1

    @property
    def dtype(self):
        return self.planes.dtype


def pack_conv_weight_h16_frag(w_oihw, dtype):
    """torch Conv2d.weight (Cout,Cin,3,3) -> FragWeightH16 (rows padded to 128, Cin to 32)"""
    _need_cuda(w_oihw)
    w = _f32c(w_oihw.detach())
    cout, cin, kh, kw = w.shape
    cout_pad, cin_pad = round_up(cout, 128), round_up(cin, 32)
    planes = torch.empty((cout_pad // 32, kh * kw * cin_pad // 32, 128, 8), device=w.device, dtype=dtype)
    check(_lib.load().sgam_pack_conv_weight_h16_frag(_p(w), _p(planes), H16[dtype], cout, cin, kh, kw, cout_pad, cin_pad, _stream()),
          "sgam_pack_conv_weight_h16_frag")
    return FragWeightH16(planes)


def _pow2_scale(maxabs):
    """power of two lifting max|w| into (512, 1024]"""
    import math
    if not (maxabs > 0) or not math.isfinite(maxabs):
        return 1.0
    return float(2.0 ** (10 - math.ceil(math.log2(maxabs))))


def split_rows(x2d, scale=1.0):
    """(N, K) fp32 view with unit inner stride -> SplitWeight planes (the B operand when it is an activation)."""
    _need_cuda(x2d)
    N, K = x2d.shape
    planes = torch.empty((round_up(N, 32) // 32, round_up(K, 32) // 32, 256, 8), device=x2d.device, dtype=torch.float16)
    check(_lib.load().sgam_split_rows_f32x(_p(x2d), _p(planes), float(scale), N, K, x2d.stride(0), _stream()),
          "sgam_split_rows_f32x")
    return SplitWeight(planes, float(scale), N, K, transient=True)


# ------------------------------------------------------------------------------------------------
# weights
# ------------------------------------------------------------------------------------------------
def pack_conv_weight(w_oihw, cout_pad=None, cin_pad=None, dtype=torch.float32):
    """torch Conv2d.weight (Cout,Cin,KH,KW) -> packed (Cout_pad, KH*KW*Cin_pad), zero padded, in `dtype`."""
    _need_cuda(w_oihw)
    w = _f32c(w_oihw.detach())
    cout, cin, kh, kw = w.shape
    # split-fp32 3x3 weights are padded to whole 128-channel tiles so that narrow convs (conv_out: 128 -> 4) run on the
    # halo-staged kernel too (one tile of mostly-zero columns costs less than the generic kernel's per-tap staging, and
    # the preceding GroupNorm fuses into it)
    # ... except outputs of at most 32 channels on the split path (the decoder's conv_out: 128 -> 4): ONE 32-channel tile, which
    # the halo kernel takes with its four wavefronts stacked along M (tile (128, 32)) — a quarter of the 128-wide tile's MFMAs
    if cout_pad is None and dtype == "f32x" and kh * kw == 9 and cout <= 32:
        cout_pad = 32
    cout_pad = cout_pad or round_up(cout, 128 if ((dtype == "f32x" or dtype in H16) and kh * kw == 9) else 64)
    cin_pad = cin_pad or round_up(cin, 32)
    if dtype == "f32x":
        planes = torch.empty((cout_pad // 32, kh * kw * cin_pad // 32, 256, 8), device=w.device, dtype=torch.float16)
        scale = _pow2_scale(float(w.abs().max()))          # once per weight version (host sync at pack time only)
        check(_lib.load().sgam_pack_conv_weight_f32x(_p(w), _p(planes), scale, cout, cin, kh, kw, cout_pad, cin_pad,
                                                     _stream()), "sgam_pack_conv_weight_f32x")
        return SplitWeight(planes, scale)
    out = torch.empty((cout_pad, kh * kw * cin_pad), device=w.device, dtype=dtype)
    if dtype == torch.float32:
        check(_lib.load().sgam_pack_conv_weight(_p(w), _p(out), cout, cin, kh, kw, cout_pad, cin_pad, _stream()),
              "sgam_pack_conv_weight")
    else:
        check(_lib.load().sgam_pack_conv_weight_h16(_p(w), _p(out), H16[dtype], cout, cin, kh, kw, cout_pad, cin_pad,
                                                    _stream()), "sgam_pack_conv_weight_h16")
    return out


# ------------------------------------------------------------------------------------------------
# conv / GEMM (MFMA implicit GEMM)
# ------------------------------------------------------------------------------------------------
# ---- autotuned launch plans (sgam_neurips22_amd/tune.py): shape key -> (bm, bn, ksplit) ----
PLAN_CACHE = {}
PLAN_RECORD = None  # set to a dict by the tuner to collect the distinct shapes of a model run
_PLAN_FILE = os.environ.get("SGAM_PLAN_FILE") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "tuned_plans_gfx950.json")


def plan_key(desc, dtype):
    return (f"{str(dtype).replace('torch.', '')}|B{desc.B}|{desc.Hi}x{desc.Wi}x{desc.Cin}|{desc.Ho}x{desc.Wo}|N{desc.N}|"
            f"k{desc.KH}x{desc.KW}s{desc.stride}u{desc.upsample2x}")


def load_plans(path=_PLAN_FILE):
    PLAN_CACHE.clear()
    if os.path.exists(path) and not os.environ.get("SGAM_NO_TUNED_PLANS"):
        import json
        with open(path) as f:
            PLAN_CACHE.update({k: tuple(v) for k, v in json.load(f).get("plans", {}).items()})
    return len(PLAN_CACHE)


def _apply_plan(desc, dtype):
    if PLAN_RECORD is not None:
        PLAN_RECORD.setdefault(plan_key(desc, dtype), None)
    if desc.plan_bm == 0 and desc.plan_ksplit == 0:
        pl = PLAN_CACHE.get(plan_key(desc, dtype))
        if pl:
            desc.plan_bm, desc.plan_bn, desc.plan_ksplit = pl


def conv_plan(desc, h16=False, split=False):
    bm, bn, ks = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
    lib = _lib.load()
    fn = lib.sgam_conv2d_f32x_plan if split else (lib.sgam_conv2d_h16_plan if h16 else lib.sgam_conv2d_plan)
    check(fn(ctypes.byref(desc), ctypes.byref(bm), ctypes.byref(bn), ctypes.byref(ks)), "sgam_conv2d_plan")
    return bm.value, bn.value, ks.value


# 16-bit AttnBlock: GroupNorm + q | k | v + fragment split as ONE launch in front of the fused attention (SGAM_ATTN_BLOCK_H16=0: the
# normalise pass, the generic 1x1 GEMM and the split launch, as before)
ATTN_BLOCK_H16 = os.environ.get("SGAM_ATTN_BLOCK_H16", "1") != "0"
ATTN_BLOCK_H16_PROJ = os.environ.get("SGAM_ATTN_BLOCK_H16_PROJ", "1") != "0"    # ... and the merge of the key ranges fused into proj_out
# split-fp32 AttnBlock as three launches (fused front end writes K / V^T in fragment order; SGAM_ATTN_BLOCK_F32X=0: q | k | v GEMM + split)
ATTN_BLOCK_F32X = os.environ.get("SGAM_ATTN_BLOCK_F32X", "1") != "0"
# the small AttnBlocks' attention (16 x 16 maps) as one launch instead of the seven of the GEMM chain (SGAM_ATTN_SMALL=0: the chain)
ATTN_SMALL = os.environ.get("SGAM_ATTN_SMALL", "1") != "0"


def _stats_out(desc, x, chunks):
    """(record tensor, chunk count) for the GroupNorm statistics of a launch that delivers `chunks` > 0 chunk records per image:
    a fresh [B][chunks][32][2] fp64 buffer"""
    if chunks <= 0:
        return None, 0
    return torch.empty((desc.B * chunks * 32 * 2,), device=x.device, dtype=torch.float64), chunks


def _run_conv(desc, x, w, bias, residual, out, gn=None, a_scale=1.0, norm=None):
    """norm = (gamma, beta, swish, groups, eps): GroupNorm(+swish) of x ahead of the product — fused into the operand
    staging where the kernel family offers it (split fp32, halo-staged 3x3), a stand-alone pass otherwise."""
    lib = _lib.load()
    split = isinstance(w, SplitWeight)
    _apply_plan(desc, "f32x" if split else x.dtype)
    if norm is not None:
        gamma, beta, swish, groups, eps = norm
        fusable = (split and FUSE_GN_APPLY and x.dtype == torch.float32 and x.dim() == 4 and groups == 32 and eps == 1e-6
                   and x.shape[3] % 128 == 0 and lib.sgam_conv2d_f32x_gn_fusable(ctypes.byref(desc)) == 1)
        # statistics that travelled with x (the producing conv's epilogue or split-K combine partials) cost one
        # 32-workgroup launch; computing them is two launches, so a map of <= 1024 pixels without them is cheaper through
        # the one-launch statistics+apply kernel
        have = getattr(x, "_gn_partials", None) is not None
        h16_fusable = (x.dtype in H16 and FUSE_GN_APPLY and x.dim() == 4 and groups == 32 and eps == 1e-6
                       and x.shape[3] % 128 == 0 and x.shape[3] <= 1024 and not desc.upsample2x and _h16_frag(w, desc) is not None)
        if fusable and have and lib.sgam_conv2d_f32x_gn_foldable(ctypes.byref(desc), x._gn_partials[1]) == 1:
            # few enough chunk partials (the group-major split-K combine of a 16^2 / 32^2 map): the conv folds them itself
            gn = (x._gn_partials, gamma, beta, swish, eps)
        elif fusable and (have or x.shape[1] * x.shape[2] > 1024):
            gn = (groupnorm_meanrstd(x, eps), gamma, beta, swish)
        elif h16_fusable and have and lib.sgam_conv2d_h16_gn_foldable(ctypes.byref(desc), x._gn_partials[1]) == 1:
            gn = (x._gn_partials, _f32c(gamma), _f32c(beta), swish, eps)      # folded inside the 16-bit halo kernel
        elif h16_fusable:
            gn = (groupnorm_meanrstd(x, eps), _f32c(gamma), _f32c(beta), swish)
        else:
            x = groupnorm_nhwc(x, gamma, beta, swish, groups, eps)
    return _run_conv_inner(lib, desc, x, w, bias, residual, out, gn, a_scale)


def _h16_frag(w, desc):
    """the fragment-ordered copy of a 16-bit 3x3 weight when this descriptor runs on the 16-bit halo kernel, else None"""
    src = getattr(w, "_sgam_frag_src", None)
    if src is None or _lib.load().sgam_conv2d_h16_uses_halo(ctypes.byref(desc)) != 1:
        return None
    fw = getattr(w, "_sgam_frag", None)
    if fw is None:
        fw = w._sgam_frag = pack_conv_weight_h16_frag(src, w.dtype)
    return fw


def _run_conv_inner(lib, desc, x, w, bias, residual, out, gn=None, a_scale=1.0):
    if isinstance(w, SplitWeight):
        if x.dtype != torch.float32:
            raise SgamHipError("split fp32 conv: fp32 activations only")
        # (opt-in, ops.PANEL_GEMM) 1x1 convolutions / plain GEMMs whose shape fits on the whole-K-panel kernel
        # (csrc/gemm_gn_f32x.hip: one barrier per 256 of K instead of one per 32, direct row stores, statistics from registers)
        M_, HW_ = desc.B * desc.Ho * desc.Wo, desc.Ho * desc.Wo
        if (PANEL_GEMM and gn is None and a_scale == 1.0 and desc.KH == 1 and desc.KW == 1 and desc.stride == 1 and not desc.upsample2x
                and desc.bias_per_row == 0 and desc.n_valid == desc.N and (M_ // 64) * (desc.N // 128) >= PANEL_MIN_WGS
                and lib.sgam_gemm_gn_f32x_fits(M_, desc.N, desc.Cin, HW_) == 1):
            cpo = desc.N // 32
            chunks = HW_ // 64 if (FUSE_GN_STATS and cpo <= 32 and cpo & (cpo - 1) == 0) else 0
            partial = torch.empty((desc.B * chunks * 32 * 2,), device=x.device, dtype=torch.float64) if chunks > 0 else None
            check(lib.sgam_gemm_panel_f32x(_p(x), desc.lda, None, None, None, _p(w.planes), float(w.scale), _p(bias), _p(residual),
                                           desc.ldr, _p(out), desc.ldc, _p(partial), M_, desc.N, desc.Cin, HW_, _stream()),
                  "sgam_gemm_panel_f32x")
            if partial is not None:
                out._gn_partials = (partial, chunks)
            return out
        ws_bytes = lib.sgam_conv2d_f32x_workspace_bytes(ctypes.byref(desc))
        if ws_bytes < 0:
            raise SgamHipError(f"sgam_conv2d_f32x: unsupported shape {[(f, getattr(desc, f)) for f, _ in desc._fields_]}")
        ws = torch.empty((ws_bytes,), device=x.device, dtype=torch.uint8) if ws_bytes else None
        # statistics of `out` for the GroupNorm that usually follows: per-chunk partial sums from the conv epilogue (no
        # split-K) or from the split-K combine (see include/sgam_hip.h)
        chunks = lib.sgam_conv2d_f32x_stats_chunks(ctypes.byref(desc)) if FUSE_GN_STATS else 0
        partial, chunks = _stats_out(desc, x, chunks)

        def tag(o):
            if partial is not None:
                o._gn_partials = (partial, chunks)
            return o

        if gn is not None:
            # GroupNorm(+swish) of x applied while the 3x3 kernel stages its input patch
            if a_scale != 1.0 or lib.sgam_conv2d_f32x_gn_fusable(ctypes.byref(desc)) != 1:
                raise SgamHipError("split fp32 conv: fused GroupNorm needs the halo-staged 3x3 kernel (sgam_conv2d_f32x_gn_fusable)")
            if len(gn) == 5:           # statistics as the producer's chunk partials, folded inside the kernel
                (part_in, chunks_in), gamma, beta, swish, eps = gn
                gamma, beta = _f32c(gamma), _f32c(beta)
                check(lib.sgam_conv2d_gnp_nhwc_f32x(ctypes.byref(desc), _p(x), _p(part_in), int(chunks_in), float(eps), _p(gamma), _p(beta),
                                                    int(swish), _p(w.planes), float(w.scale), _p(bias), _p(residual), _p(out), _p(partial),
                                                    _p(ws), ws_bytes, _stream()), "sgam_conv2d_gnp_nhwc_f32x")
                return tag(out)
            mean_rstd, gamma, beta, swish = gn
            gamma, beta = _f32c(gamma), _f32c(beta)
            check(lib.sgam_conv2d_gn_nhwc_f32x(ctypes.byref(desc), _p(x), _p(mean_rstd), _p(gamma), _p(beta), int(swish),
                                               _p(w.planes), float(w.scale), _p(bias), _p(residual), _p(out), _p(partial),
                                               _p(ws), ws_bytes, _stream()), "sgam_conv2d_gn_nhwc_f32x")
            return tag(out)
        if partial is not None:
            check(lib.sgam_conv2d_stats_nhwc_f32x(ctypes.byref(desc), _p(x), float(a_scale), _p(w.planes), float(w.scale),
                                                  _p(bias), _p(residual), _p(out), _p(partial), _p(ws), ws_bytes, _stream()),
                  "sgam_conv2d_stats_nhwc_f32x")
            return tag(out)
        check(lib.sgam_conv2d_nhwc_f32x(ctypes.byref(desc), _p(x), float(a_scale), _p(w.planes), float(w.scale), _p(bias),
                                        _p(residual), _p(out), _p(ws), ws_bytes, _stream()), "sgam_conv2d_nhwc_f32x")
        return out
    if x.dtype in H16:
        if w.dtype != x.dtype or (residual is not None and residual.dtype != x.dtype):
            raise SgamHipError("16-bit conv: operands must share one 16-bit dtype")
        fw = _h16_frag(w, desc)
        if fw is not None:
            # halo-staged 3x3 kernel of the 16-bit mode (csrc/h16_halo.hip): optional GroupNorm(+swish) of x while staging,
            # statistics of `out` from the epilogue
            chunks = lib.sgam_conv2d_h16_stats_chunks(ctypes.byref(desc)) if (FUSE_GN_STATS and out.dtype in H16) else 0
            partial, chunks = _stats_out(desc, x, chunks)
            ws_bytes = lib.sgam_conv2d_halo_h16_workspace_bytes(ctypes.byref(desc))
            ws = torch.empty((ws_bytes,), device=x.device, dtype=torch.uint8) if ws_bytes > 0 else None
            if gn is not None and len(gn) == 5:          # statistics as the producer's chunk partials, folded inside the kernel
                (part_in, chunks_in), gamma, beta, swish, eps = gn
                check(lib.sgam_conv2d_halo_gnp_nhwc_h16(ctypes.byref(desc), H16[x.dtype], _p(x), _p(part_in), int(chunks_in), float(eps),
                                                        _p(gamma), _p(beta), int(swish), _p(fw.planes), _p(bias), _p(residual), _p(out),
                                                        int(out.dtype == torch.float32), _p(partial), _p(ws), max(ws_bytes, 0),
                                                        _stream()), "sgam_conv2d_halo_gnp_nhwc_h16")
                if partial is not None:
                    out._gn_partials = (partial, chunks)
                return out
            mr, gamma, beta, swish = gn if gn is not None else (None, None, None, False)
            check(lib.sgam_conv2d_halo_nhwc_h16(ctypes.byref(desc), H16[x.dtype], _p(x), _p(mr), _p(gamma), _p(beta), int(swish),
                                                _p(fw.planes), _p(bias), _p(residual), _p(out), int(out.dtype == torch.float32),
                                                _p(partial), _p(ws), max(ws_bytes, 0), _stream()), "sgam_conv2d_halo_nhwc_h16")
            if partial is not None:
                out._gn_partials = (partial, chunks)
            return out
        if gn is not None:
            raise SgamHipError("16-bit conv: fused GroupNorm needs the halo-staged 3x3 kernel (sgam_conv2d_h16_uses_halo)")
        ws_bytes = lib.sgam_conv2d_h16_workspace_bytes(ctypes.byref(desc))
        if ws_bytes < 0:
            raise SgamHipError(f"sgam_conv2d_h16: unsupported shape {[(f, getattr(desc, f)) for f, _ in desc._fields_]}")
        ws = torch.empty((ws_bytes,), device=x.device, dtype=torch.uint8) if ws_bytes else None
        # statistics of `out` for the GroupNorm that usually follows (1x1 / strided convolutions, proj_out): per-chunk partial
        # sums from the direct epilogue of the generic kernel
        chunks = lib.sgam_conv2d_h16_generic_stats_chunks(ctypes.byref(desc)) if (FUSE_GN_STATS and out.dtype in H16) else 0
        if chunks > 0:
            partial, chunks = _stats_out(desc, x, chunks)
            check(lib.sgam_conv2d_stats_nhwc_h16(ctypes.byref(desc), H16[x.dtype], _p(x), _p(w), _p(bias), _p(residual), _p(out), 0,
                                                 _p(partial), _p(ws), ws_bytes, _stream()), "sgam_conv2d_stats_nhwc_h16")
            out._gn_partials = (partial, chunks)
            return out
        check(lib.sgam_conv2d_nhwc_h16(ctypes.byref(desc), H16[x.dtype], _p(x), _p(w), _p(bias), _p(residual), _p(out),
                                       int(out.dtype == torch.float32), _p(ws), ws_bytes, _stream()),
              "sgam_conv2d_nhwc_h16")
        return out
    ws_bytes = lib.sgam_conv2d_workspace_bytes(ctypes.byref(desc))
    if ws_bytes < 0:
        raise SgamHipError(f"sgam_conv2d: unsupported shape {[(f, getattr(desc, f)) for f, _ in desc._fields_]}")
    ws = torch.empty((ws_bytes,), device=x.device, dtype=torch.uint8) if ws_bytes else None
    table, swish = gn if gn is not None else (None, False)
    check(lib.sgam_conv2d_gn_nhwc_f32(ctypes.byref(desc), _p(x), _p(table), int(swish), _p(w), _p(bias), _p(residual),
                                      _p(out), _p(ws), ws_bytes, _stream()), "sgam_conv2d_gn_nhwc_f32")
    return out


def conv2d_nhwc(x, w_packed, bias, *, cout, kh, kw, stride=1, pad_t=0, pad_l=0, pad_b=None, pad_r=None,
                upsample2x=False, residual=None, cin=None, gn=None, out_dtype=None, norm=None):
    """x (B,Hi,Wi,Cx) NHWC fp32 -> (B,Ho,Wo,cout).  w_packed from pack_conv_weight (rows padded to 64,
    Cin padded to 32).  `cin` = channels of x actually contracted (defaults to Cx, must be % 32)."""
    _need_cuda(x, w_packed)
    _dense_nhwc(x, "conv2d_nhwc")
    B, Hi, Wi, Cx = x.shape
    cin = cin or Cx
    pad_b = pad_t if pad_b is None else pad_b
    pad_r = pad_l if pad_r is None else pad_r
    Hl, Wl = (2 * Hi, 2 * Wi) if upsample2x else (Hi, Wi)
    Ho = (Hl + pad_t + pad_b - kh) // stride + 1
    Wo = (Wl + pad_l + pad_r - kw) // stride + 1
    N = w_packed.shape[0]
    out = torch.empty((B, Ho, Wo, cout), device=x.device, dtype=out_dtype or x.dtype)
    d = ConvDesc(B=B, Hi=Hi, Wi=Wi, Cin=cin, Ho=Ho, Wo=Wo, N=N, KH=kh, KW=kw, stride=stride, pad_t=pad_t,
                 pad_l=pad_l, upsample2x=int(upsample2x), lda=x.stride(2), ldb=w_packed.stride(0), ldc=cout,
                 ldr=(residual.stride(2) if residual is not None else 0), n_valid=cout, bias_per_row=0)
    return _run_conv(d, x, w_packed, bias, residual, out, gn, norm=norm)


def gemm_nt(a, b, bias=None, residual=None, bias_per_row=False, out=None, gn=None, out_dtype=None, a_scale=1.0):
    """out[M][N] = a[M][K] @ b[N][K]^T (+bias) (+residual).  a, b: 2-D fp32 CUDA tensors with unit
    inner stride (row strides free, so column slices of a fused projection can be passed directly).
    K % 32 == 0, N % 4 == 0."""
    _need_cuda(a)
    if a.dtype == torch.float32 and F32_MODE == "split" and gn is None and not isinstance(b, SplitWeight):
        b = split_rows(b)       # activation x activation product (q k^T, P v): split the B side on the fly
    M, K = a.shape
    N, K2 = b.shape
    assert K == K2 and a.stride(1) == 1 and b.stride(1) == 1
    if out is None:
        out = torch.empty((M, N), device=a.device, dtype=out_dtype or a.dtype)
    d = ConvDesc(B=1, Hi=1, Wi=M, Cin=K, Ho=1, Wo=M, N=N, KH=1, KW=1, stride=1, pad_t=0, pad_l=0, upsample2x=0,
                 lda=a.stride(0), ldb=b.stride(0), ldc=out.stride(0),
                 ldr=(residual.stride(0) if residual is not None else 0), n_valid=N,
                 bias_per_row=int(bias_per_row))
    return _run_conv(d, a, b, bias, residual, out, gn, a_scale)


def gemm_gn_fits(M, N, K, HW):
    return F32_MODE == "split" and _lib.load().sgam_gemm_gn_f32x_fits(M, N, K, HW) == 1


def gemm_gn_f32x(x2d, mean_rstd, gamma, beta, w, bias, hw):
    """out[M][N] = GroupNorm(x)[M][K] @ W^T + bias with the normalisation fused into the operand staging (AttnBlock's q | k | v
    projection on the split-fp32 path; csrc/gemm_gn_f32x.hip).  x2d (M, K) fp32 rows of NHWC pixels, `hw` rows per image,
    mean_rstd (B, 32, 2), w a SplitWeight of the stacked (N, K) weights."""
    _need_cuda(x2d)
    M, K = x2d.shape
    N = w.shape[0]
    out = torch.empty((M, N), device=x2d.device, dtype=torch.float32)
    check(_lib.load().sgam_gemm_gn_f32x(_p(x2d), x2d.stride(0), _p(mean_rstd), _p(_f32c(gamma)), _p(_f32c(beta)), _p(w.planes),
                                        float(w.scale), _p(bias), _p(out), N, M, N, K, hw, _stream()), "sgam_gemm_gn_f32x")
    return out


# ------------------------------------------------------------------------------------------------
# GroupNorm(+swish), softmax
# ------------------------------------------------------------------------------------------------
def groupnorm_nhwc(x, gamma, beta, swish, groups=32, eps=1e-6):
    _need_cuda(x, gamma, beta)
    B, H, W, C = x.shape
    lib = _lib.load()
    if x.dtype in H16:
        ws_bytes = lib.sgam_groupnorm_h16_workspace_bytes(B, H * W, C)
        if ws_bytes < 0:
            raise SgamHipError(f"sgam_groupnorm_h16: unsupported shape B={B} HW={H * W} C={C}")
        ws = torch.empty((ws_bytes,), device=x.device, dtype=torch.uint8)
        y = torch.empty_like(x)
        pre = getattr(x, "_gn_partials", None)
        if pre is not None and H * W > 1024 and groups == 32:
            partial, chunks = pre     # statistics came with the tensor (16-bit halo conv epilogue): finalize + apply only
            gamma, beta = _f32c(gamma), _f32c(beta)
            check(lib.sgam_groupnorm_from_partials_h16(_p(x), _p(partial), chunks, _p(gamma), _p(beta), _p(y), H16[x.dtype], B,
                                                       H * W, C, groups, eps, int(swish), _p(ws), ws_bytes, _stream()),
                  "sgam_groupnorm_from_partials_h16")
            return y
        check(lib.sgam_groupnorm_nhwc_h16(_p(x), _p(gamma), _p(beta), _p(y), H16[x.dtype], B, H * W, C, groups, eps,
                                          int(swish), _p(ws), ws_bytes, _stream()), "sgam_groupnorm_nhwc_h16")
        return y
    ws_bytes = lib.sgam_groupnorm_workspace_bytes(B, H * W, C)
    if ws_bytes < 0:
        raise SgamHipError(f"sgam_groupnorm: unsupported shape B={B} HW={H * W} C={C}")
    ws = torch.empty((ws_bytes,), device=x.device, dtype=torch.uint8)
    y = torch.empty_like(x)
    pre = getattr(x, "_gn_partials", None)
    if pre is not None and H * W > 1024 and groups == 32:
        partial, chunks = pre     # statistics came with the tensor (conv epilogue): finalize + apply only
        check(lib.sgam_groupnorm_from_partials_f32(_p(x), _p(partial), chunks, _p(gamma), _p(beta), _p(y), B, H * W, C, groups,
                                                   eps, int(swish), _p(ws), ws_bytes, _stream()),
              "sgam_groupnorm_from_partials_f32")
        return y
    check(lib.sgam_groupnorm_nhwc_f32(_p(x), _p(gamma), _p(beta), _p(y), B, H * W, C, groups, eps, int(swish),
                                      _p(ws), ws_bytes, _stream()), "sgam_groupnorm_nhwc_f32")
    return y


def groupnorm_stats(x, gamma, beta, groups=32, eps=1e-6):
    """Statistics-only GroupNorm: returns the (B, C, 2) {scale, shift} table consumed by the fused conv prologue
    (`gn=(table, swish)` of conv2d_nhwc / gemm_nt)."""
    _need_cuda(x, gamma, beta)
    B, H, W, C = x.shape
    lib = _lib.load()
    table = torch.empty((B, C, 2), device=x.device, dtype=torch.float32)
    ws_bytes = lib.sgam_groupnorm_workspace_bytes(B, H * W, C)
    if ws_bytes < 0:
        raise SgamHipError(f"sgam_groupnorm: unsupported shape B={B} HW={H * W} C={C}")
    ws = torch.empty((ws_bytes,), device=x.device, dtype=torch.uint8)
    check(lib.sgam_groupnorm_stats_nhwc_f32(_p(x), _p(gamma), _p(beta), _p(table), B, H * W, C, groups, eps, _p(ws),
                                            ws_bytes, _stream()), "sgam_groupnorm_stats_nhwc_f32")
    return table


def groupnorm_meanrstd(x, eps=1e-6):
    """(B, 32, 2) {mean, rstd} per (image, group) of an NHWC fp32 tensor, for convolutions that normalise their input
    while staging it: one 32-workgroup launch when the conv that produced x left its partial sums (`_gn_partials`), else
    a statistics pass + the fold."""
    _need_cuda(x)
    B, H, W, C = x.shape
    lib = _lib.load()
    out = torch.empty((B, 32, 2), device=x.device, dtype=torch.float32)
    pre = getattr(x, "_gn_partials", None)
    if pre is not None:
        partial, chunks = pre
        check(lib.sgam_groupnorm_stats_from_partials_f32(_p(partial), chunks, _p(out), B, H * W, C, 32, eps, _stream()),
              "sgam_groupnorm_stats_from_partials_f32")
        return out
    ws_bytes = lib.sgam_groupnorm_workspace_bytes(B, H * W, C)
    if ws_bytes < 0:
        raise SgamHipError(f"sgam_groupnorm: unsupported shape B={B} HW={H * W} C={C}")
    ws = torch.empty((ws_bytes,), device=x.device, dtype=torch.uint8)
    if x.dtype in H16:
        check(lib.sgam_groupnorm_meanrstd_nhwc_h16(_p(x), _p(out), H16[x.dtype], B, H * W, C, 32, eps, _p(ws), ws_bytes, _stream()),
              "sgam_groupnorm_meanrstd_nhwc_h16")
        return out
    check(lib.sgam_groupnorm_meanrstd_nhwc_f32(_p(x), _p(out), B, H * W, C, 32, eps, _p(ws), ws_bytes, _stream()),
          "sgam_groupnorm_meanrstd_nhwc_f32")
    return out


def softmax_rows_h16(s, scale, dtype, block=0):
    """fp32 scores (rows, cols) -> 16-bit probabilities softmax(scale * s); block > 0: block-diagonal (softmax_rows_)."""
    _need_cuda(s)
    rows, cols = s.shape
    p = torch.empty((rows, cols), device=s.device, dtype=dtype)
    if block:
        check(_lib.load().sgam_softmax_rows_blockdiag_h16(_p(s), _p(p), H16[dtype], rows, cols, s.stride(0), p.stride(0), float(scale),
                                                          int(block), _stream()), "sgam_softmax_rows_blockdiag_h16")
        return p
    check(_lib.load().sgam_softmax_rows_h16(_p(s), _p(p), H16[dtype], rows, cols, s.stride(0), p.stride(0), float(scale),
                                            _stream()), "sgam_softmax_rows_h16")
    return p


def transpose_h16(x2d):
    """(HW, C) 16-bit view with unit inner stride -> contiguous (C, HW)."""
    _need_cuda(x2d)
    HW, C = x2d.shape
    y = torch.empty((C, HW), device=x2d.device, dtype=x2d.dtype)
    check(_lib.load().sgam_transpose_h16(_p(x2d), _p(y), H16[x2d.dtype], C, HW, x2d.stride(0), _stream()),
          "sgam_transpose_h16")
    return y


FUSED_ATTENTION = os.environ.get("SGAM_FUSED_ATTN", "1") != "0"


def attention_fusable(n, C):
    """True when sgam_attention_f32x takes the shape (C = 256, n a multiple of 256)."""
    return FUSED_ATTENTION and _lib.load().sgam_attention_f32x_workspace_bytes(int(n), int(C)) > 0 and n % 256 == 0


def attention(qkv, C, scale, out=None, B=1):
    """softmax(q k^T * scale) v for the fused projection qkv = [q | k | v] (B * n, 3C) fp32 — B images of n tokens stacked along
    the rows, every query attending to the keys of its own image — in one pass over the keys (csrc/attention.hip): the (n, n)
    score matrix is never written, and a batch is ONE launch sequence."""
    _need_cuda(qkv)
    nt = qkv.shape[0]
    assert qkv.dtype == torch.float32 and qkv.shape[1] == 3 * C and qkv.stride(1) == 1 and nt % B == 0
    n = nt // B
    lib = _lib.load()
    ws_bytes = lib.sgam_attention_f32x_batched_workspace_bytes(n, C, B)
    if ws_bytes < 0:
        raise SgamHipError(f"sgam_attention_f32x: unsupported shape n={n} C={C} B={B}")
    ws = torch.empty((ws_bytes,), device=qkv.device, dtype=torch.uint8)
    if out is None:
        out = torch.empty((nt, C), device=qkv.device, dtype=torch.float32)
    check(lib.sgam_attention_f32x_batched(_p(qkv), _p(qkv[:, C:]), _p(qkv[:, 2 * C:]), qkv.stride(0), n, C, B, float(scale), _p(out),
                                          out.stride(0), _p(ws), ws_bytes, _stream()), "sgam_attention_f32x_batched")
    return out


# AttnBlock.proj_out inside the attention call (csrc/attention.hip: attn_combine_proj_f32x_kernel): the merge of the key ranges becomes
# the operand staging of the projection — one launch and a [n][C] round trip fewer per block.  SGAM_ATTN_PROJ=0: separate launches.
ATTN_PROJ = os.environ.get("SGAM_ATTN_PROJ", "1") == "1"


def attention_proj(qkv, C, scale, wp, bias, residual, out=None, B=1):
    """`attention` followed by proj_out (+ bias, + residual) as one launch sequence: qkv (B * n, 3C) fp32, wp the SplitWeight of
    proj_out's (C, C) weight, residual / out (B * n, C).  The result carries the GroupNorm chunk statistics of the block output
    (`_gn_partials`, one chunk per 32-row tile) like every convolution's."""
    _need_cuda(qkv)
    nt = qkv.shape[0]
    assert qkv.dtype == torch.float32 and qkv.shape[1] == 3 * C and qkv.stride(1) == 1 and nt % B == 0
    assert isinstance(wp, SplitWeight)
    n = nt // B
    lib = _lib.load()
    ws_bytes = lib.sgam_attention_f32x_batched_workspace_bytes(n, C, B)
    if ws_bytes < 0:
        raise SgamHipError(f"sgam_attention_proj_f32x: unsupported shape n={n} C={C} B={B}")
    ws = torch.empty((ws_bytes,), device=qkv.device, dtype=torch.uint8)
    if out is None:
        out = torch.empty((nt, C), device=qkv.device, dtype=torch.float32)
    assert out.stride(1) == 1 and (residual is None or (residual.stride(1) == 1 and residual.dtype == torch.float32))
    chunks = n // 32
    partial = torch.empty((B * chunks * 32 * 2,), device=qkv.device, dtype=torch.float64) if FUSE_GN_STATS else None
    check(lib.sgam_attention_proj_f32x_batched(_p(qkv), _p(qkv[:, C:]), _p(qkv[:, 2 * C:]), qkv.stride(0), n, C, B, float(scale),
                                               _p(wp.planes), float(wp.scale), _p(bias), _p(residual),
                                               residual.stride(0) if residual is not None else 0, _p(out), out.stride(0), _p(partial),
                                               _p(ws), ws_bytes, _stream()), "sgam_attention_proj_f32x_batched")
    if partial is not None:
        out._gn_partials = (partial, chunks)
    return out


# channel <-> MFMA-row permutation of the transposed products (attention.hip): row 8 j + 4 h + i of a 32-row tile carries channel 16 h + 4 j + i
_TILE_CHANNEL_OF_ROW = [16 * ((r >> 2) & 1) + 4 * (r >> 3) + (r & 3) for r in range(32)]


def permute_rows_for_transposed_product(w2d):
    """(N, K) weight with the rows of every 32-row tile re-ordered for a kernel that computes the product transposed and wants a
    lane's sixteen accumulator slots to be sixteen consecutive channels"""
    N = w2d.shape[0]
    assert N % 32 == 0
    idx = (torch.arange(N, device=w2d.device).view(N // 32, 32)[:, torch.tensor(_TILE_CHANNEL_OF_ROW, device=w2d.device)]).reshape(-1)
    return w2d.index_select(0, idx).contiguous()


def attn_block_f32x(x2d, mean_rstd, gamma, beta, wqkv_perm, bqkv, C, scale, wp, bp, B=1, out=None):
    """The whole AttnBlock of the split-fp32 path (reference diffusionmodules/model.py:168-192) in three launches: fused front end
    (GroupNorm + q | k | v, K / V^T straight in fragment order), one-pass attention, merge + proj_out + residual x.  wqkv_perm: the
    SplitWeight of permute_rows_for_transposed_product(stacked weight); statistics of the output travel as `_gn_partials`."""
    _need_cuda(x2d)
    nt = x2d.shape[0]
    assert x2d.dtype == torch.float32 and x2d.shape[1] == C and x2d.stride(1) == 1 and nt % B == 0
    assert isinstance(wp, SplitWeight) and isinstance(wqkv_perm, SplitWeight)
    n = nt // B
    lib = _lib.load()
    ws_bytes = lib.sgam_attn_block_f32x_workspace_bytes(n, C, B)
    if ws_bytes < 0:
        raise SgamHipError(f"sgam_attn_block_f32x: unsupported shape n={n} C={C} B={B}")
    ws = torch.empty((ws_bytes,), device=x2d.device, dtype=torch.uint8)
    if out is None:
        out = torch.empty((nt, C), device=x2d.device, dtype=torch.float32)
    chunks = n // 32
    partial = torch.empty((B * chunks * 32 * 2,), device=x2d.device, dtype=torch.float64) if FUSE_GN_STATS else None
    check(lib.sgam_attn_block_f32x(_p(x2d), x2d.stride(0), _p(mean_rstd), _p(_f32c(gamma)), _p(_f32c(beta)), _p(wqkv_perm.planes),
                                   float(wqkv_perm.scale), _p(bqkv), n, C, B, float(scale), _p(wp.planes), float(wp.scale), _p(bp),
                                   _p(out), out.stride(0), _p(partial), _p(ws), ws_bytes, _stream()), "sgam_attn_block_f32x")
    if partial is not None:
        out._gn_partials = (partial, chunks)
    return out


def attention_small_fits(n, C, B=1):
    return ATTN_SMALL and _lib.load().sgam_attention_small_f32x_fits(int(n), int(C), int(B)) == 1


def attention_small(qkv, C, scale, B=1, out=None):
    """softmax(q k^T scale) v of the small AttnBlocks (n = 256 tokens per image, C = 512) in ONE launch: a workgroup holds a query tile's
    whole score row (csrc/attention.hip: attn_small_f32x_kernel); B images stacked along the rows attend within themselves."""
    _need_cuda(qkv)
    nt = qkv.shape[0]
    assert qkv.shape[1] == 3 * C and qkv.stride(1) == 1 and nt % B == 0
    if out is None:
        out = torch.empty((nt, C), device=qkv.device, dtype=qkv.dtype)
    if qkv.dtype in H16:
        check(_lib.load().sgam_attention_small_h16(_p(qkv), _p(qkv[:, C:]), _p(qkv[:, 2 * C:]), H16[qkv.dtype], qkv.stride(0), nt // B, C, B,
                                                   float(scale), _p(out), out.stride(0), _stream()), "sgam_attention_small_h16")
        return out
    assert qkv.dtype == torch.float32
    check(_lib.load().sgam_attention_small_f32x(_p(qkv), _p(qkv[:, C:]), _p(qkv[:, 2 * C:]), qkv.stride(0), nt // B, C, B, float(scale),
                                                _p(out), out.stride(0), _stream()), "sgam_attention_small_f32x")
    return out


def attention_h16(qkv, C, scale, out=None, B=1):
    """16-bit throughput variant of `attention` (qkv bf16 / fp16, result in the same dtype)."""
    _need_cuda(qkv)
    nt = qkv.shape[0]
    assert qkv.dtype in H16 and qkv.shape[1] == 3 * C and qkv.stride(1) == 1 and nt % B == 0
    n = nt // B
    lib = _lib.load()
    ws_bytes = lib.sgam_attention_h16_batched_workspace_bytes(n, C, B)
    if ws_bytes < 0:
        raise SgamHipError(f"sgam_attention_h16: unsupported shape n={n} C={C} B={B}")
    ws = torch.empty((ws_bytes,), device=qkv.device, dtype=torch.uint8)
    if out is None:
        out = torch.empty((nt, C), device=qkv.device, dtype=qkv.dtype)
    check(lib.sgam_attention_h16_batched(_p(qkv), _p(qkv[:, C:]), _p(qkv[:, 2 * C:]), H16[qkv.dtype], qkv.stride(0), n, C, B,
                                         float(scale), _p(out), out.stride(0), _p(ws), ws_bytes, _stream()), "sgam_attention_h16_batched")
    return out


def pack_qkv_weight_h16(w32, dtype):
    """stacked q | k | v weight (3C, C) fp32 -> the fragment order of the fused 16-bit AttnBlock front end (attn_block_h16)"""
    _need_cuda(w32)
    C = w32.shape[1]
    assert w32.dtype == torch.float32 and w32.shape[0] == 3 * C and w32.is_contiguous()
    out = torch.empty((3 * C * C,), device=w32.device, dtype=dtype)
    check(_lib.load().sgam_pack_qkv_weight_h16(_p(w32), _p(out), H16[dtype], C, _stream()), "sgam_pack_qkv_weight_h16")
    return out


def attn_block_h16_fusable(x, n, C, B):
    """the block input is 16-bit, carries its producer's chunk statistics and has the fused attention's shape"""
    pre = getattr(x, "_gn_partials", None)
    return (ATTN_BLOCK_H16 and x.dtype in H16 and pre is not None and pre[0].dtype == torch.float64 and pre[1] > 0
            and _lib.load().sgam_attn_block_h16_workspace_bytes(n, C, B) > 0)


def pack_weight_tp_h16(w2d, dtype):
    """(rows, C) fp32 1x1 weight -> transposed-product fragment order (attn_block_h16's proj_out operand)"""
    _need_cuda(w2d)
    rows, C = w2d.shape
    assert w2d.dtype == torch.float32 and w2d.is_contiguous()
    out = torch.empty((rows * C,), device=w2d.device, dtype=dtype)
    check(_lib.load().sgam_pack_weight_tp_h16(_p(w2d), _p(out), H16[dtype], rows, C, _stream()), "sgam_pack_weight_tp_h16")
    return out


def attn_block_h16(x2d, pre, gamma, beta, eps, w_frag, bias, C, scale, B=1, out=None, proj=None):
    """16-bit AttnBlock ahead of proj_out: GroupNorm (from the producer's chunk statistics `pre` = (partials, chunks)) applied inside
    the q | k | v projection, K / V^T written in the attention's fragment order, flash + merge: four launches (seven unfused)."""
    _need_cuda(x2d)
    nt = x2d.shape[0]
    assert x2d.dtype in H16 and x2d.shape[1] == C and x2d.stride(1) == 1 and nt % B == 0
    n = nt // B
    lib = _lib.load()
    ws_bytes = lib.sgam_attn_block_h16_workspace_bytes(n, C, B)
    if ws_bytes < 0:
        raise SgamHipError(f"sgam_attn_block_h16: unsupported shape n={n} C={C} B={B}")
    ws = torch.empty((ws_bytes,), device=x2d.device, dtype=torch.uint8)
    if out is None:
        out = torch.empty((nt, C), device=x2d.device, dtype=x2d.dtype)
    partial, chunks = pre
    gamma, beta, bias = _f32c(gamma), _f32c(beta), _f32c(bias)      # (the kernel reads fp32: a .half() / .bfloat16() model is converted here)
    if proj is not None:
        # proj = (wp_frag, bp): the merge of the key ranges fused into proj_out + residual x — the whole block, with the chunk statistics
        # of what it stored for the GroupNorm that follows
        wp_frag, bp = proj
        bp = _f32c(bp)
        po = torch.empty((B * (n // 32) * 32 * 2,), device=x2d.device, dtype=torch.float64) if FUSE_GN_STATS else None
        check(lib.sgam_attn_block_proj_h16(_p(x2d), x2d.stride(0), _p(partial), int(chunks), _p(gamma), _p(beta), float(eps), _p(w_frag),
                                           _p(bias), H16[x2d.dtype], n, C, B, float(scale), _p(wp_frag), _p(bp), _p(po), _p(out),
                                           out.stride(0), _p(ws), ws_bytes, _stream()), "sgam_attn_block_proj_h16")
        if po is not None:
            out._gn_partials = (po, n // 32)
        return out
    check(lib.sgam_attn_block_h16(_p(x2d), x2d.stride(0), _p(partial), int(chunks), _p(gamma), _p(beta), float(eps), _p(w_frag), _p(bias),
                                  H16[x2d.dtype], n, C, B, float(scale), _p(out), out.stride(0), _p(ws), ws_bytes, _stream()),
          "sgam_attn_block_h16")
    return out


def softmax_rows_(s, scale, block=0):
    """in place: s = softmax(scale * s) over each row; block > 0: every row over the columns of ITS diagonal block only, exact
    zeros elsewhere (B images' scores as one matrix: a query never attends to another image's keys)."""
    _need_cuda(s)
    rows, cols = s.shape
    if block:
        check(_lib.load().sgam_softmax_rows_blockdiag_f32(_p(s), rows, cols, s.stride(0), float(scale), int(block), _stream()),
              "sgam_softmax_rows_blockdiag_f32")
        return s
    check(_lib.load().sgam_softmax_rows_f32(_p(s), rows, cols, s.stride(0), float(scale), _stream()),
          "sgam_softmax_rows_f32")
    return s


# ------------------------------------------------------------------------------------------------
# warps and frame feedback
# ------------------------------------------------------------------------------------------------
# which forward splat runs: False (default) = the two-pass device-scope atomicMax form, True (SGAM_SPLAT_TILED=1) = target-owned LDS
# z-tiles (no global atomics), None ("auto") = the tiled form from SPLAT_TILED_MIN_POINTS source points on.  Measured (DESIGN.md
# 4.2, bench `roofline_warp`): on the scene loop's own geometry (smooth depth: neighbouring points hit neighbouring pixels, the
# atomics coalesce) the two-pass form is 10 % FASTER at every size (150 against 167 us for 12.6 M points); on an incoherent scatter
# (white-noise depth) the tiled form is 1.9 x faster (183 against 352 us).  Both are bit-identical to the reference.
SPLAT_TILED = {"0": False, "1": True, "auto": None}.get(os.environ.get("SGAM_SPLAT_TILED", "0"), False)
SPLAT_TILED_MIN_POINTS = int(os.environ.get("SGAM_SPLAT_TILED_MIN_POINTS", 1 << 20))
_SPLAT_WS = {}


def _splat_workspace(dev, B, N, H, W, out=None):
    """scratch of the tiled splat (tile bitmaps, cached target pixels), zero-filled ONCE at allocation (the kernels leave the
    bitmaps zero): one buffer per (device, STREAM, shape), cached; (None, 0, None) when the two-pass form is to run.  The stream is
    part of the key because pass 2 of a call reads (and clears) what its pass 1 wrote: two scenes of the same shape on two HIP
    streams (`distributed.ConcurrentScenes`) sharing one buffer would overwrite each other's target pixels and bitmaps between
    the passes — wrong winners, no error."""
    use = SPLAT_TILED if SPLAT_TILED is not None else (B * N * H * W >= SPLAT_TILED_MIN_POINTS)
    if not use:
        return None, 0, None
    nb = _lib.load().sgam_forward_splat_workspace_bytes(B, N, H, W)
    if nb < 0:
        return None, 0, None
    key = (str(dev), int(torch.cuda.current_stream(dev).cuda_stream), B, N, H, W)
    if key not in _SPLAT_WS:
        if len(_SPLAT_WS) > 64:
            _SPLAT_WS.clear()
        _SPLAT_WS[key] = torch.zeros((nb,), device=dev, dtype=torch.uint8)
    return _SPLAT_WS[key], nb, key


def _splat_tiled_check(rc, what, key):
    """`check` for the tiled splat: a launch that failed between the two passes may leave registered bins behind, and the next
    call on this buffer relies on zero bitmaps — the buffer is dropped (the next call allocates a zero-filled one)"""
    if rc != 0:
        _SPLAT_WS.pop(key, None)
    check(rc, what)


def forward_splat(src_feats, src_depths, tgt_K, src_Kinv, T, *, channels_last=False, depth_range=None,
                  dataset=None, want=("merge_depths", "merge_feats", "extrap"), extrap_bool=False):
    """Forward splat + median fill + mask (+ normalised x).  src_feats (B,N,3,H,W) or, with
    channels_last, (B,N,H,W,3); src_depths (B,N,H,W); tgt_K (B,3,3); src_Kinv (B*N,3,3); T (B*N,4,4).
    Returns a dict with the tensors named in `want` (any of merge_depths, merge_feats, extrap, x,
    proj_feats, proj_depth, inb_mask, pix_xy).  extrap is uint8 0/1, or torch.bool (same bytes) with extrap_bool."""
    _need_cuda(src_feats, src_depths, tgt_K, src_Kinv, T)
    f = _f32c(src_feats)
    d = _f32c(src_depths)
    B, N, H, W = d.shape
    HW = H * W
    dev = d.device
    cs, ps = (1, 3) if channels_last else (HW, 1)
    o = {}
    mk = lambda shape, dt=torch.float32: torch.empty(shape, device=dev, dtype=dt)  # noqa: E731
    if "merge_depths" in want: o["merge_depths"] = mk((B, 1, H, W))
    if "merge_feats" in want: o["merge_feats"] = mk((B, 3, H, W))
    if "extrap" in want: o["extrap"] = mk((B, 1, H, W), torch.bool if extrap_bool else torch.uint8)
    if "x" in want: o["x"] = mk((B, 4, H, W))
    if "proj_feats" in want: o["proj_feats"] = mk((B, 3, H, W))
    if "proj_depth" in want: o["proj_depth"] = mk((B, 1, H, W))
    if "inb_mask" in want: o["inb_mask"] = mk((B * HW * N,), torch.uint8)
    if "pix_xy" in want: o["pix_xy"] = mk((B * HW * N, 2), torch.int32)
    dr = None
    if depth_range is not None:
        dr = (ctypes.c_float * 2)(float(depth_range[0]), float(depth_range[1]))
    norm = DATASET_NORM.get(dataset, 0) if dataset else 0
    if "x" in want and norm == 0:
        raise NotImplementedError(f"dataset {dataset!r}")
    # keep every contiguous copy alive in a local until the launch is enqueued: a temporary passed as
    # `_p(_f32c(t))` is freed right after `_p` returns, and the caching allocator may hand its block to the NEXT
    # temporary, whose copy kernel then lands before ours on the same stream
    kt, kinv, tt = _f32c(tgt_K), _f32c(src_Kinv), _f32c(T)
    ws, nb, wkey = _splat_workspace(dev, B, N, H, W) if ("inb_mask" not in want and "pix_xy" not in want) else (None, 0, None)
    if ws is not None:
        _splat_tiled_check(_lib.load().sgam_forward_splat_tiled_f32(
            _p(f), cs, ps, _p(d), _p(kt), _p(kinv), _p(tt), B, N, H, W, dr, norm, _p(ws), nb,
            _p(o.get("merge_depths")), _p(o.get("merge_feats")), _p(o.get("extrap")), _p(o.get("x")),
            _p(o.get("proj_feats")), _p(o.get("proj_depth")), _stream()), "sgam_forward_splat_tiled_f32", wkey)
        return o
    winner = mk((B, HW), torch.int32)
    check(_lib.load().sgam_forward_splat_f32(
        _p(f), cs, ps, _p(d), _p(kt), _p(kinv), _p(tt), B, N, H, W, dr, norm, _p(winner),
        _p(o.get("merge_depths")), _p(o.get("merge_feats")), _p(o.get("extrap")), _p(o.get("x")),
        _p(o.get("proj_feats")), _p(o.get("proj_depth")), _p(o.get("inb_mask")), _p(o.get("pix_xy")), _stream()),
        "sgam_forward_splat_f32")
    return o


def _ptr_table(tensors):
    return (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


def forward_splat_srcs(src_feats, src_depths, tgt_K, src_Kinv, T, *, B=1, depth_range=None, dataset=None,
                       want=("x", "extrap"), extrap_bool=False, out=None):
    """`forward_splat` with the sources as a LIST of B*N per-frame tensors — features (H,W,3) channels-last fp32
    contiguous and depths (H,W) — read in place through a pointer table (no stacked copy).  `out`: optional dict of
    preallocated outputs (persistent buffers of the scene loop)."""
    n_src = len(src_feats)
    assert n_src == len(src_depths) and n_src % B == 0 and 0 < n_src <= 64
    N = n_src // B
    for f, d in zip(src_feats, src_depths):
        _need_cuda(f, d)
        if f.dtype != torch.float32 or d.dtype != torch.float32 or not f.is_contiguous() or not d.is_contiguous():
            raise SgamHipError("forward_splat_srcs: sources must be contiguous fp32 (H,W,3) / (H,W) tensors")
    H, W = src_depths[0].shape[-2:]
    HW = H * W
    dev = src_depths[0].device
    out = out or {}
    o = {}
    mk = lambda name, shape, dt=torch.float32: out[name] if name in out else torch.empty(shape, device=dev, dtype=dt)  # noqa: E731
    if "merge_depths" in want: o["merge_depths"] = mk("merge_depths", (B, 1, H, W))
    if "merge_feats" in want: o["merge_feats"] = mk("merge_feats", (B, 3, H, W))
    if "extrap" in want: o["extrap"] = mk("extrap", (B, 1, H, W), torch.bool if extrap_bool else torch.uint8)
    if "x" in want: o["x"] = mk("x", (B, 4, H, W))
    dr = None
    if depth_range is not None:
        dr = (ctypes.c_float * 2)(float(depth_range[0]), float(depth_range[1]))
    norm = DATASET_NORM.get(dataset, 0) if dataset else 0
    if "x" in want and norm == 0:
        raise NotImplementedError(f"dataset {dataset!r}")
    kt, kinv, tt = _f32c(tgt_K), _f32c(src_Kinv), _f32c(T)       # locals keep any contiguous copy alive until enqueued
    ws, nb, wkey = _splat_workspace(dev, B, N, H, W, out)
    if ws is not None:
        _splat_tiled_check(_lib.load().sgam_forward_splat_tiled_srcs_f32(
            _ptr_table(src_feats), _ptr_table(src_depths), 1, 3, _p(kt), _p(kinv), _p(tt), B, N, H, W, dr, norm, _p(ws), nb,
            _p(o.get("merge_depths")), _p(o.get("merge_feats")), _p(o.get("extrap")), _p(o.get("x")), None, None, _stream()),
            "sgam_forward_splat_tiled_srcs_f32", wkey)
        return o
    winner = mk("winner", (B, HW), torch.int32)
    check(_lib.load().sgam_forward_splat_srcs_f32(
        _ptr_table(src_feats), _ptr_table(src_depths), 1, 3, _p(kt), _p(kinv), _p(tt), B, N, H, W, dr, norm, _p(winner),
        _p(o.get("merge_depths")), _p(o.get("merge_feats")), _p(o.get("extrap")), _p(o.get("x")), None, None, None, None,
        _stream()), "sgam_forward_splat_srcs_f32")
    return o


def inverse_warp_srcs(src_imgs, src_depths, tgt_depth, src_K, tgt_Kinv, T_tgt2src, B=1, out=None):
    """`inverse_warp` over a LIST of B*N per-frame sources: images (H,W,3) channels-last, depths (H,W), in place.
    `out`: optional contiguous (B,3,H,W) destination (e.g. the rgb planes of a persistent B = 1 model input)."""
    n_src = len(src_imgs)
    assert n_src == len(src_depths) and n_src % B == 0 and 0 < n_src <= 64
    for f, d in zip(src_imgs, src_depths):
        _need_cuda(f, d)
        if f.dtype != torch.float32 or d.dtype != torch.float32 or not f.is_contiguous() or not d.is_contiguous():
            raise SgamHipError("inverse_warp_srcs: sources must be contiguous fp32 (H,W,3) / (H,W) tensors")
    H, W = src_depths[0].shape[-2:]
    if out is None:
        out = torch.empty((B, 3, H, W), device=src_depths[0].device, dtype=torch.float32)
    assert out.shape == (B, 3, H, W) and out.is_contiguous() and out.dtype == torch.float32
    td, sk, tk, tt = _f32c(tgt_depth), _f32c(src_K), _f32c(tgt_Kinv), _f32c(T_tgt2src)  # keep alive
    check(_lib.load().sgam_inverse_warp_srcs_f32(_ptr_table(src_imgs), _ptr_table(src_depths), 1, 3, _p(td), _p(sk), _p(tk),
                                                 _p(tt), B, n_src // B, H, W, _p(out), None, _stream()),
          "sgam_inverse_warp_srcs_f32")
    return out


def inverse_warp(src_imgs, src_depths, tgt_depth, src_K, tgt_Kinv, T_tgt2src, want_zbuf=False):
    _need_cuda(src_imgs, src_depths, tgt_depth, src_K, tgt_Kinv, T_tgt2src)
    im = _f32c(src_imgs)
    B, N, _, H, W = im.shape
    out = torch.empty((B, 3, H, W), device=im.device, dtype=torch.float32)
    zb = torch.empty((B, H, W), device=im.device, dtype=torch.float32) if want_zbuf else None
    sd, td, sk, tk, tt = _f32c(src_depths), _f32c(tgt_depth), _f32c(src_K), _f32c(tgt_Kinv), _f32c(T_tgt2src)  # keep alive
    check(_lib.load().sgam_inverse_warp_f32(_p(im), _p(sd), _p(td), _p(sk), _p(tk), _p(tt), B, N, H, W, _p(out), _p(zb),
                                            _stream()), "sgam_inverse_warp_f32")
    return (out, zb) if want_zbuf else out


from .ops_aux import *        # noqa: F401,F403,E402 (kernel timeline, layout hops, vector quantiser)
from .ops_aux import _kernel_name  # noqa: F401,E402

load_plans()
