"""Plumbing shared by the tensor-level wrappers (ops.py, ops_aux.py): stream / pointer helpers, layout checks, dtype codes and the
fp32 <-> 16-bit cast.  torch is used only for device allocations, the current HIP stream and `data_ptr()`."""
import ctypes

import torch

from . import _lib
from ._lib import SgamHipError, check

GE, CLEVR = 1, 2
DATASET_NORM = {"google_earth": GE, "clevr-infinite": CLEVR}


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise SgamHipError("sgam_neurips22_amd runs on the HIP backend only: got a CPU tensor "
                               "(move the model and inputs to 'cuda'; there is no CPU fallback)")


def _f32c(t):
    if t.dtype != torch.float32:
        t = t.float()
    return t if t.is_contiguous() else t.contiguous()


def _dense_nhwc(x, what):
    """(B,H,W,C) with free channel pitch but dense pixel / row / image pitches — the only layout the kernels address"""
    B, H, W, _ = x.shape
    ld = x.stride(2)
    if x.stride(3) != 1 or (W > 1 and ld < x.shape[3]) or (H > 1 and x.stride(1) != W * ld) or \
            (B > 1 and x.stride(0) != H * W * ld):
        raise SgamHipError(f"{what}: NHWC tensor with strides {tuple(x.stride())} is not pixel-dense "
                           "(call .contiguous() on slices taken along a leading axis)")


def round_up(v, m):
    return (v + m - 1) // m * m


# 16-bit throughput path: `ht` code of the C ABI per torch dtype
H16 = {torch.bfloat16: 0, torch.float16: 1}
DTYPES = {"f32": torch.float32, "fp32": torch.float32, "bf16": torch.bfloat16, "fp16": torch.float16,
          "f16": torch.float16}


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


def cast(x, dtype):
    """fp32 <-> 16-bit conversion on the GPU (RNE)."""
    _need_cuda(x)
    if x.dtype == dtype:
        return x
    x = _c(x)
    y = torch.empty(x.shape, device=x.device, dtype=dtype)
    lib = _lib.load()
    if x.dtype == torch.float32 and dtype in H16:
        check(lib.sgam_cast_f32_h16(_p(x), _p(y), H16[dtype], x.numel(), _stream()), "sgam_cast_f32_h16")
    elif x.dtype in H16 and dtype == torch.float32:
        check(lib.sgam_cast_h16_f32(_p(x), _p(y), H16[x.dtype], x.numel(), _stream()), "sgam_cast_h16_f32")
    else:
        raise SgamHipError(f"cast {x.dtype} -> {dtype} not supported")
    return y
