"""Tensor-level wrappers over the C ABI that are not convolutions / attention (those live in ops.py, which re-exports everything
here): the per-kernel timeline (measurement), the NCHW <-> NHWC hops and the conditional model's 5 -> 4 channel head, the vector
quantiser, the depth codec and the frame feedback."""
import ctypes

import torch

from . import _lib
from ._lib import SgamHipError, check
from ._opscore import *        # noqa: F401,F403
from ._opscore import _c, _dense_nhwc, _f32c, _need_cuda, _p, _stream  # noqa: F401

# ------------------------------------------------------------------------------------------------
# kernel timeline (measurement only)
# ------------------------------------------------------------------------------------------------
def _kernel_name(kernel, where):
    """launch-site spelling -> `name<args>` with symbolic template arguments resolved from the enclosing function"""
    import re
    name = kernel.strip().strip("()").replace(" ", "")
    m = re.search(r"\[([^\]]*)\]\s*$", where or "")
    if m and "<" in name:
        sub = dict(kv.split(" = ") for kv in m.group(1).split(", ") if " = " in kv)
        head, args = name.split("<", 1)
        args = [sub.get(a, a) for a in args.rstrip(">").split(",")]
        name = f"{head}<{','.join(args)}>"
    return name


def kernel_timeline(fn, empty_brackets=32):
    """Run fn() with the library's per-kernel HIP-event brackets on (include/sgam_hip.h, sgam_prof_*): returns
    (records, bracket_ms) — records = [(kernel name, ms, flops, bytes, (M, N, K, ksplit))] in launch order, elapsed times as measured (the
    caller subtracts bracket_ms, the median cost of a bracket around nothing).  Eager launches only (no graph replay)."""
    lib = _lib.load()
    torch.cuda.synchronize()
    check(lib.sgam_prof_enable(1), "sgam_prof_enable")
    try:
        # park the GPU behind a spin so that the host enqueues everything ahead of it: launches then run back to back
        # and a bracket is the kernel's duration, not the host's launch gap
        torch.cuda._sleep(int(1.0e8))
        fn()
        for _ in range(empty_brackets):
            lib.sgam_prof_mark_empty(_stream())
        torch.cuda.synchronize()
    finally:
        lib.sgam_prof_enable(0)
    recs, empties = [], []
    k, w = ctypes.c_char_p(), ctypes.c_char_p()
    ms, fl, by = ctypes.c_float(), ctypes.c_double(), ctypes.c_double()
    shp = (ctypes.c_int32 * 4)()
    for i in range(lib.sgam_prof_count()):
        check(lib.sgam_prof_get(i, ctypes.byref(k), ctypes.byref(w), ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(by)),
              "sgam_prof_get")
        if k.value == b"(empty)":
            empties.append(ms.value)
        else:
            lib.sgam_prof_get_shape(i, shp)
            recs.append((_kernel_name(k.value.decode(), (w.value or b"").decode()), ms.value, fl.value, by.value, tuple(shp)))
    empties.sort()
    return recs, (empties[len(empties) // 2] if empties else 0.0)


# ------------------------------------------------------------------------------------------------
# layout hops
# ------------------------------------------------------------------------------------------------
def nchw_to_nhwc(x, c_pad=None):
    _need_cuda(x)
    x = _f32c(x)
    B, C, H, W = x.shape
    ld = c_pad or C
    y = (torch.zeros if ld != C else torch.empty)((B, H, W, ld), device=x.device, dtype=torch.float32)
    check(_lib.load().sgam_nchw_to_nhwc_f32(_p(x), _p(y), B, C, H * W, ld, _stream()), "sgam_nchw_to_nhwc_f32")
    return y


def nhwc_to_nchw(x, c=None):
    _need_cuda(x)
    _dense_nhwc(x, "nhwc_to_nchw")
    B, H, W, ld = x.shape
    c = c or ld
    y = torch.empty((B, c, H, W), device=x.device, dtype=torch.float32)
    check(_lib.load().sgam_nhwc_to_nchw_f32(_p(x), _p(y), B, c, H * W, x.stride(2), _stream()),
          "sgam_nhwc_to_nchw_f32")
    return y


def encode_head(x_nchw, mask, w, bias, ld=32, dtype=torch.float32):
    """cat(x, mask) -> conv1x1(5->4) -> NHWC (B,H,W,ld) with channels 4.. zero (model.py:107-113)."""
    _need_cuda(x_nchw, w, bias)
    x = _f32c(x_nchw)
    B, C, H, W = x.shape
    assert C == 4
    m = None
    if mask is not None:
        m = mask.reshape(B, H * W)
        if m.dtype == torch.bool and m.is_contiguous():
            m = m.view(torch.uint8)          # same 0/1 bytes: no conversion kernel inside the step
        else:
            m = (m != 0).to(torch.uint8).contiguous()
    y = torch.empty((B, H, W, ld), device=x.device, dtype=dtype)
    wf, bf = _f32c(w.detach().reshape(4, 5)), _f32c(bias.detach())
    if dtype == torch.float32:
        check(_lib.load().sgam_encode_head_f32(_p(x), _p(m), _p(wf), _p(bf), _p(y), B, H * W, ld, _stream()),
              "sgam_encode_head_f32")
    else:
        check(_lib.load().sgam_encode_head_h16(_p(x), _p(m), _p(wf), _p(bf), _p(y), H16[dtype], B, H * W, ld, _stream()),
              "sgam_encode_head_h16")
    return y


# ------------------------------------------------------------------------------------------------
# vector quantiser
# ------------------------------------------------------------------------------------------------
def row_sumsq(x):
    _need_cuda(x)
    x = _f32c(x.detach())
    out = torch.empty((x.shape[0],), device=x.device, dtype=torch.float32)
    check(_lib.load().sgam_row_sumsq_f32(_p(x), _p(out), x.shape[0], x.shape[1], _stream()), "sgam_row_sumsq_f32")
    return out


def vq_nearest(z_tokens, codebook, e_sq, straight_through=True, want_dist=False, want_zq=True):
    """z_tokens (T,D) -> (idx int64 (T,), z_q (T,D) or None, dist (T,n_e) or None)."""
    _need_cuda(z_tokens, codebook, e_sq)
    z = _f32c(z_tokens)
    T, D = z.shape
    n_e = codebook.shape[0]
    lib = _lib.load()
    dots = torch.empty((T, n_e), device=z.device, dtype=torch.float32)
    idx = torch.empty((T,), device=z.device, dtype=torch.int64)
    zq = torch.empty((T, D), device=z.device, dtype=torch.float32) if want_zq else None
    dist = torch.empty((T, n_e), device=z.device, dtype=torch.float32) if want_dist else None
    ws_bytes = lib.sgam_vq_workspace_bytes(T, D, n_e)
    if ws_bytes < 0:
        raise SgamHipError(f"sgam_vq: unsupported shape T={T} D={D} n_e={n_e}")
    ws = torch.empty((ws_bytes,), device=z.device, dtype=torch.uint8) if ws_bytes else None
    check(lib.sgam_vq_nearest_f32(_p(z), _p(codebook), _p(e_sq), _p(dots), _p(idx), _p(zq), _p(dist), T, D, n_e,
                                  int(straight_through), _p(ws), ws_bytes, _stream()), "sgam_vq_nearest_f32")
    return idx, zq, dist


def vq_commit_loss(z_tokens, codebook, idx, beta):
    """scalar commitment loss mean((e[idx]-z)^2) + beta*mean((e[idx]-z)^2) (quantize.py:296-301) as a 0-d tensor"""
    _need_cuda(z_tokens, codebook, idx)
    z = _f32c(z_tokens)
    T, D = z.shape
    partial = torch.empty((T,), device=z.device, dtype=torch.float64)
    loss = torch.empty((1,), device=z.device, dtype=torch.float32)
    idx = idx.reshape(-1)
    check(_lib.load().sgam_vq_commit_loss_f32(_p(z), _p(codebook), _p(idx), _p(partial), _p(loss), T, D, codebook.shape[0],
                                              float(beta), _stream()), "sgam_vq_commit_loss_f32")
    return loss[0]


def vq_gather(codebook, idx):
    _need_cuda(codebook, idx)
    idx = idx.reshape(-1).to(torch.int64).contiguous()
    T, D = idx.numel(), codebook.shape[1]
    out = torch.empty((T, D), device=codebook.device, dtype=torch.float32)
    check(_lib.load().sgam_vq_gather_f32(_p(codebook), _p(idx), _p(out), T, D, codebook.shape[0], _stream()),
          "sgam_vq_gather_f32")
    return out


def vq_topk(dist, k):
    _need_cuda(dist)
    T, n_e = dist.shape
    vals = torch.empty((T, k), device=dist.device, dtype=torch.float32)
    inds = torch.empty((T, k), device=dist.device, dtype=torch.int64)
    check(_lib.load().sgam_vq_topk_f32(_p(dist), _p(vals), _p(inds), T, n_e, k, _stream()), "sgam_vq_topk_f32")
    return vals, inds


# ------------------------------------------------------------------------------------------------
# depth codec and frame feedback
# ------------------------------------------------------------------------------------------------
def depth_normalise(depth, dataset, compute_mask=True, mask_bool=False, out=None, out_mask=None):
    """depth (any shape) -> (normalised inverse depth, extrap mask uint8 (torch.bool with mask_bool: the same 0/1
    bytes) or None)  [model.py:196-229].  `out` / `out_mask`: optional contiguous destinations of the same size."""
    _need_cuda(depth)
    d = _f32c(depth)
    if out is None:
        out = torch.empty_like(d)
    assert out.numel() == d.numel() and out.is_contiguous() and out.dtype == torch.float32
    em = None
    if compute_mask:
        em = out_mask if out_mask is not None else torch.empty(d.shape, device=d.device,
                                                               dtype=torch.bool if mask_bool else torch.uint8)
        assert em.numel() == d.numel() and em.is_contiguous() and em.element_size() == 1
    if dataset not in DATASET_NORM:
        raise NotImplementedError(f"dataset {dataset!r}")
    check(_lib.load().sgam_depth_normalise_f32(_p(d), int(compute_mask), _p(em), _p(out), DATASET_NORM[dataset],
                                               d.numel(), _stream()), "sgam_depth_normalise_f32")
    return out, em


_LUT = {}


def rgb_lut(device):
    """lut[u] = float32(float64(u) / 127.5 - 1.0): what prepare_batch_data reads back from the uint8 PNG
    (inference_pipeline.py:534 then the .astype(np.float32) at :607)."""
    key = str(device)
    if key not in _LUT:
        import numpy as np
        lut = (np.arange(256, dtype=np.float64) / 127.5 - 1.0).astype(np.float32)
        _LUT[key] = torch.from_numpy(lut).to(device)
    return _LUT[key]


def rgb_u8_to_f32(u8):
    """uint8 image -> fp32 lut[u8] (the PNG re-read of prepare_batch_data, inference_pipeline.py:534)"""
    _need_cuda(u8)
    u8 = _c(u8)
    assert u8.dtype == torch.uint8
    out = torch.empty(u8.shape, device=u8.device, dtype=torch.float32)
    check(_lib.load().sgam_rgb_u8_to_f32(_p(u8), _p(rgb_lut(u8.device)), _p(out), u8.numel(), _stream()),
          "sgam_rgb_u8_to_f32")
    return out


def frame_feedback(dec, dataset, want_u8=False):
    """dec (B,4,H,W) -> (rgb_f (B,H,W,3) fp32, depth (B,H,W) fp32[, rgb_u8 (B,H,W,3) uint8])."""
    _need_cuda(dec)
    dec = _f32c(dec)
    B, C, H, W = dec.shape
    assert C == 4
    rgb_f = torch.empty((B, H, W, 3), device=dec.device, dtype=torch.float32)
    depth = torch.empty((B, H, W), device=dec.device, dtype=torch.float32)
    u8 = torch.empty((B, H, W, 3), device=dec.device, dtype=torch.uint8) if want_u8 else None
    check(_lib.load().sgam_frame_feedback_f32(_p(dec), _p(rgb_lut(dec.device)), DATASET_NORM[dataset], _p(u8), _p(rgb_f),
                                              _p(depth), B, H * W, _stream()), "sgam_frame_feedback_f32")
    return (rgb_f, depth, u8) if want_u8 else (rgb_f, depth)
