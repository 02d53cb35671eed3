"""Encoder / Decoder of the conditional VQGAN on the HIP backend.

Same constructor kwargs, attribute names and ``state_dict`` keys as the reference's
``sgam/generative_sensing_module/modules/diffusionmodules/model.py`` (Encoder :342-433, Decoder
:437-539, ResnetBlock :78-137, AttnBlock :140-192, Upsample :38-53, Downsample :56-75, Normalize
:34-35), so checkpoints and callers drop in unchanged — but no layer ever runs a torch op: the
``nn.Conv2d`` / ``nn.GroupNorm`` objects are parameter containers only, and every ``forward`` is a
sequence of calls into libsgam_hip.so (MFMA implicit-GEMM convolutions, GroupNorm+swish, softmax).

Activations stay NHWC between layers (``forward_nhwc``); the public ``forward`` of each module takes
and returns NCHW like the reference and pays one layout hop on each side.
"""
import numpy as np
import os

import torch
import torch.nn as nn

from .... import ops


class Conv2d(nn.Conv2d):
    """nn.Conv2d as a parameter container + HIP execution.  Packed weights ([Cout_pad][taps][Cin_pad],
    K contiguous — the B operand layout of the implicit GEMM) are cached per (storage, version)."""

    def _packed(self, dtype=torch.float32):
        w = self.weight
        key = (w.data_ptr(), w._version, str(w.device))
        if getattr(self, "_pack_key", None) != key:
            self._packs = {}
            self._pack_bias = None if self.bias is None else self.bias.detach().float().contiguous()
            self._pack_key = key
        if dtype == torch.float32 and ops.F32_MODE == "split":
            dtype = "f32x"      # fp32 activations, weights pre-split into fp16 hi/lo planes (conv_f32x.hip)
        if dtype not in self._packs:
            self._packs[dtype] = ops.pack_conv_weight(w, dtype=dtype)
            if dtype in ops.H16 and self.kernel_size == (3, 3) and self.stride == (1, 1):
                # the 16-bit halo kernel reads a fragment-ordered copy (packed on first use, csrc/h16_halo.hip)
                self._packs[dtype]._sgam_frag_src = w
        return self._packs[dtype], self._pack_bias

    def forward_nhwc(self, x, residual=None, upsample2x=False, pad=None, gn=None, out_dtype=None, norm=None):
        """gn = (scale/shift table from GroupNorm.stats_nhwc, swish flag): GroupNorm(+swish) of the input fused
        into the operand staging of the implicit GEMM (fp32 path only).  The kernel family follows x.dtype:
        fp32 -> fp32-in MFMA parity path, bf16/fp16 -> 16-bit MFMA throughput path."""
        wp, b = self._packed(x.dtype)
        kh, kw = self.kernel_size
        if pad is None:
            pad = (self.padding[0], self.padding[1], self.padding[0], self.padding[1])  # t, l, b, r
        return ops.conv2d_nhwc(x, wp, b, cout=self.out_channels, kh=kh, kw=kw, stride=self.stride[0],
                               pad_t=pad[0], pad_l=pad[1], pad_b=pad[2], pad_r=pad[3], upsample2x=upsample2x,
                               residual=residual, cin=wp.shape[1] // (kh * kw), gn=gn, out_dtype=out_dtype, norm=norm)

    def forward(self, x):
        cin_pad = self._packed()[0].shape[1] // (self.kernel_size[0] * self.kernel_size[1])
        return ops.nhwc_to_nchw(self.forward_nhwc(ops.nchw_to_nhwc(x, c_pad=cin_pad)))


class GroupNorm(nn.GroupNorm):
    def stats_nhwc(self, x):
        """(B,C,2) scale/shift table for the fused conv prologue."""
        return ops.groupnorm_stats(x, self.weight.detach(), self.bias.detach(), groups=self.num_groups, eps=self.eps)

    def forward_nhwc(self, x, swish=False):
        return ops.groupnorm_nhwc(x, self.weight.detach(), self.bias.detach(), swish, groups=self.num_groups,
                                  eps=self.eps)

    def forward(self, x):
        return ops.nhwc_to_nchw(self.forward_nhwc(ops.nchw_to_nhwc(x)))


# Measured on MI355X (profiles/): with fp32-in MFMA the convolution is matrix-pipe bound and the
# per-tap re-normalisation of the fused prologue (9x the swish work, on the same wavefronts that feed the MFMAs)
# costs more (+19 us on the 128->128 @256^2 layer) than the stand-alone HBM-bound normalise pass it removes
# (~12 us).  The fused kernel stays available (and parity-tested) behind this switch.
FUSE_GROUPNORM_INTO_CONV = False
# AttnBlock on the split-fp32 path: normalise inside the q | k | v GEMM (measured)
FUSE_NORM_INTO_QKV = os.environ.get("SGAM_FUSE_NORM_QKV", "1") != "0"


def _norm_conv(norm, swish, conv, x, **kw):
    """GroupNorm(+swish) followed by a convolution."""
    if FUSE_GROUPNORM_INTO_CONV:
        return conv.forward_nhwc(x, gn=(norm.stats_nhwc(x), swish), **kw)
    if (x.dtype == torch.float32 and ops.F32_MODE == "split") or x.dtype in ops.H16:
        # ops decides per launch: normalise inside the halo-staged 3x3 kernel where that kernel runs, else a separate pass
        return conv.forward_nhwc(x, norm=(norm.weight.detach(), norm.bias.detach(), swish, norm.num_groups, norm.eps), **kw)
    return conv.forward_nhwc(norm.forward_nhwc(x, swish=swish), **kw)


def Normalize(in_channels):
    return GroupNorm(num_groups=32, num_channels=in_channels, eps=1e-6, affine=True)


def nonlinearity(x):
    """swish; standalone NCHW use only — inside the network swish is fused into the GroupNorm kernel."""
    raise ops.SgamHipError("nonlinearity() is fused into GroupNorm on the HIP backend (GroupNorm.forward_nhwc(swish=True))")


class _NHWCModule(nn.Module):
    def forward(self, x, *unused):
        return ops.nhwc_to_nchw(self.forward_nhwc(ops.nchw_to_nhwc(x)))


class Upsample(_NHWCModule):
    def __init__(self, in_channels, with_conv):
        super().__init__()
        self.with_conv = with_conv
        if not with_conv:
            raise NotImplementedError("Upsample(with_conv=False) is not on the SGAM hot path")
        self.conv = Conv2d(in_channels, in_channels, kernel_size=3, stride=1, padding=1)

    def forward_nhwc(self, x):
        # nearest 2x folded into the conv's gather (K6 + K1)
        return self.conv.forward_nhwc(x, upsample2x=True)


class Downsample(_NHWCModule):
    def __init__(self, in_channels, with_conv):
        super().__init__()
        self.with_conv = with_conv
        if not with_conv:
            raise NotImplementedError("Downsample(with_conv=False) is not on the SGAM hot path")
        self.conv = Conv2d(in_channels, in_channels, kernel_size=3, stride=2, padding=0)

    def forward_nhwc(self, x):
        # F.pad(x, (0,1,0,1)) + stride-2 conv: zero padding on the right/bottom only (K2)
        return self.conv.forward_nhwc(x, pad=(0, 0, 1, 1))


class ResnetBlock(_NHWCModule):
    def __init__(self, *, in_channels, out_channels=None, conv_shortcut=False, dropout=0.0, temb_channels=512):
        super().__init__()
        out_channels = in_channels if out_channels is None else out_channels
        self.in_channels, self.out_channels = in_channels, out_channels
        self.use_conv_shortcut = conv_shortcut
        self.norm1 = Normalize(in_channels)
        self.conv1 = Conv2d(in_channels, out_channels, kernel_size=3, stride=1, padding=1)
        if temb_channels > 0:
            self.temb_proj = nn.Linear(temb_channels, out_channels)  # never used: temb is None on this path
        self.norm2 = Normalize(out_channels)
        self.dropout = nn.Dropout(dropout)  # p = 0 in every SGAM config; identity at inference
        self.conv2 = Conv2d(out_channels, out_channels, kernel_size=3, stride=1, padding=1)
        if in_channels != out_channels:
            if conv_shortcut:
                self.conv_shortcut = Conv2d(in_channels, out_channels, kernel_size=3, stride=1, padding=1)
            else:
                self.nin_shortcut = Conv2d(in_channels, out_channels, kernel_size=1, stride=1, padding=0)

    def forward_nhwc(self, x):
        h = _norm_conv(self.norm1, True, self.conv1, x)
        if self.in_channels != self.out_channels:
            x = (self.conv_shortcut if self.use_conv_shortcut else self.nin_shortcut).forward_nhwc(x)
        return _norm_conv(self.norm2, True, self.conv2, h, residual=x)  # x + h in the conv epilogue

    def forward(self, x, temb=None):
        if temb is not None:
            raise NotImplementedError("timestep embeddings are not used by SGAM's VQGAN")
        return super().forward(x)


BLOCKDIAG_MAX_ROWS = 4096      # B * n up to which a batch of small attention blocks runs as one block-diagonal chain (64 MB of scores)


class AttnBlock(_NHWCModule):
    """Single-head spatial self-attention (reference :168-192) as MFMA GEMMs + a row softmax:
    [q|k|v] = GN(x) Wqkv^T (GroupNorm fused into the operand staging), S = q k^T, P = softmax(S c^-1/2),
    O = P v (v transposed once so that the key axis is contiguous), out = x + O Wp^T."""

    def __init__(self, in_channels):
        super().__init__()
        self.in_channels = in_channels
        self.norm = Normalize(in_channels)
        self.q = Conv2d(in_channels, in_channels, kernel_size=1, stride=1, padding=0)
        self.k = Conv2d(in_channels, in_channels, kernel_size=1, stride=1, padding=0)
        self.v = Conv2d(in_channels, in_channels, kernel_size=1, stride=1, padding=0)
        self.proj_out = Conv2d(in_channels, in_channels, kernel_size=1, stride=1, padding=0)

    def _packed_qkv(self):
        ws = (self.q.weight, self.k.weight, self.v.weight)
        key = tuple((w.data_ptr(), w._version) for w in ws) + (str(ws[0].device),)
        if getattr(self, "_qkv_key", None) != key:
            c = self.in_channels
            self._wqkv = {torch.float32: torch.cat([m.weight.detach().reshape(c, c) for m in (self.q, self.k, self.v)],
                                                   0).float().contiguous()}
            self._bqkv = torch.cat([m.bias.detach() for m in (self.q, self.k, self.v)]).float().contiguous()
            self._qkv_key = key
        return self._wqkv, self._bqkv

    def forward_nhwc(self, x):
        B, H, W, C = x.shape
        n = H * W
        wqkvs, bqkv = self._packed_qkv()
        wkey = "f32x" if (x.dtype == torch.float32 and ops.F32_MODE == "split" and not FUSE_GROUPNORM_INTO_CONV) else x.dtype
        if wkey not in wqkvs:
            w32 = wqkvs[torch.float32]
            wqkvs[wkey] = ops.split_rows(w32, ops._pow2_scale(float(w32.abs().max()))) if wkey == "f32x" else ops.cast(w32, x.dtype)
        wqkv = wqkvs[wkey]
        wp, bp = self.proj_out._packed(x.dtype)
        if x.dtype in ops.H16:
            return self._forward_nhwc_h16(x, wqkv, bqkv, wp, bp)
        fused_qkv = wkey == "f32x" and FUSE_NORM_INTO_QKV and ops.gemm_gn_fits(B * n, 3 * C, C, n)
        if fused_qkv and ops.ATTN_BLOCK_F32X and ops.ATTN_PROJ and ops.attention_fusable(n, C) and isinstance(wp, ops.SplitWeight):
            # the whole block in three launches: GroupNorm + q | k | v with K / V^T written straight in the attention's fragment order
            # (the GEMM + split launch's arithmetic, equal to fp32 round-off), one pass over the keys, merge + proj_out + residual (csrc/attention.hip)
            if "f32x_perm" not in wqkvs:
                w32 = wqkvs[torch.float32]
                wqkvs["f32x_perm"] = ops.split_rows(ops.permute_rows_for_transposed_product(w32), wqkv.scale)
            mr = ops.groupnorm_meanrstd(x, self.norm.eps)
            ob = ops.attn_block_f32x(x.reshape(B * n, C), mr, self.norm.weight.detach(), self.norm.bias.detach(), wqkvs["f32x_perm"], bqkv,
                                     C, int(C) ** (-0.5), wp, bp, B=B)
            out = ob.view(B, H, W, C)
            if hasattr(ob, "_gn_partials"):
                out._gn_partials = ob._gn_partials
            return out
        if fused_qkv:
            # GroupNorm applied while the q | k | v GEMM stages its operand (csrc/gemm_gn_f32x.hip): no normalise pass
            mr = ops.groupnorm_meanrstd(x, self.norm.eps)
            qkv_all = ops.gemm_gn_f32x(x.reshape(B * n, C), mr, self.norm.weight.detach(), self.norm.bias.detach(), wqkv, bqkv, n)
            table, h = None, None
        elif FUSE_GROUPNORM_INTO_CONV:
            table, h = self.norm.stats_nhwc(x), x                         # (B, C, 2), no swish for attention
        else:
            table, h = None, self.norm.forward_nhwc(x, swish=False)
        scale = int(C) ** (-0.5)
        if B > 1 and fused_qkv and ops.attention_fusable(n, C):
            # a batch (lock-stepped scenes, warp candidates) is ONE launch sequence: the images are stacked along the rows of
            # the q | k | v projection, the fused attention keeps every query inside its image, and proj_out runs as the 1x1
            # convolution it is (per-image GroupNorm statistics of the block output from its epilogue)
            if ops.ATTN_PROJ and isinstance(wp, ops.SplitWeight):
                ob = ops.attention_proj(qkv_all, C, scale, wp, bp, x.reshape(B * n, C), B=B)
                out = ob.view(B, H, W, C)
                if hasattr(ob, "_gn_partials"):
                    out._gn_partials = ob._gn_partials       # per-image chunk statistics of the block output
                return out
            o = ops.attention(qkv_all, C, scale, B=B)
            return self.proj_out.forward_nhwc(o.view(B, H, W, C), residual=x)
        if fused_qkv and ops.F32_MODE == "split" and ops.attention_small_fits(n, C, B):
            # the 16 x 16 blocks: scores, soft-max and P v of a query tile in ONE launch (the chain below is seven), any batch
            o = ops.attention_small(qkv_all, C, scale, B=B)
            return self.proj_out.forward_nhwc(o.view(B, H, W, C), residual=x)
        if B > 1 and fused_qkv and B * n <= BLOCKDIAG_MAX_ROWS and n % 4 == 0:
            # the small blocks of a batch (16 x 16 maps, C = 512: not the fused kernel's shape) as ONE block-diagonal chain: a
            # (B n) x (B n) score matrix whose soft-max keeps a query inside its image — 8 x the score FLOPs of B separate
            # chains, all of 4 GF at B = 8, against 5 launches per IMAGE
            vt = ops.nhwc_to_nchw(qkv_all[:, 2 * C:].unsqueeze(0).unsqueeze(0), c=C).view(C, B * n)
            s = ops.gemm_nt(qkv_all[:, :C], qkv_all[:, C:2 * C])
            ops.softmax_rows_(s, scale, block=n)
            o = ops.gemm_nt(s, vt, a_scale=1024.0)
            return self.proj_out.forward_nhwc(o.view(B, H, W, C), residual=x)
        out = torch.empty_like(x)
        for b in range(B):
            xb = x[b].reshape(n, C)
            qkv = qkv_all[b * n:(b + 1) * n] if fused_qkv else ops.gemm_nt(
                h[b].reshape(n, C), wqkv, bias=bqkv, gn=None if table is None else (table[b:b + 1], False))   # (n, 3C)
            if ops.F32_MODE == "split" and ops.attention_fusable(n, C) and ops.ATTN_PROJ and isinstance(wp, ops.SplitWeight):
                # one pass over the keys AND proj_out + residual: the merge of the key ranges is the projection's operand staging
                ob = ops.attention_proj(qkv, C, scale, wp, bp, xb, out=out[b].reshape(n, C))
                if B == 1 and hasattr(ob, "_gn_partials"):
                    out._gn_partials = ob._gn_partials
                continue
            if ops.F32_MODE == "split" and ops.attention_fusable(n, C):
                o = ops.attention(qkv, C, scale)                           # one pass over the keys, no (n, n) scores
            else:
                vt = ops.nhwc_to_nchw(qkv[:, 2 * C:].unsqueeze(0).unsqueeze(0), c=C).view(C, n)   # (C, n) = v^T
                s = ops.gemm_nt(qkv[:, :C], qkv[:, C:2 * C])               # (n, n) scores
                ops.softmax_rows_(s, scale)
                o = ops.gemm_nt(s, vt, a_scale=1024.0)                     # (n, C); probabilities lifted before the split
            ob = ops.gemm_nt(o, wp, bias=bp, residual=xb, out=out[b].reshape(n, C))
            if B == 1 and hasattr(ob, "_gn_partials"):
                out._gn_partials = ob._gn_partials   # statistics of the block output for the next GroupNorm
        return out


def _attn_h16(self, x, wqkv, bqkv, wp, bp):
    """16-bit attention: q/k/v and P in bf16/fp16, scores and softmax in fp32."""
    B, H, W, C = x.shape
    n = H * W
    scale = int(C) ** (-0.5)
    if ops.attention_fusable(n, C) and ops.attn_block_h16_fusable(x, n, C, B):
        # GroupNorm inside the q | k | v projection, K / V^T straight into the attention's fragment order: 4 launches ahead of proj_out
        # (csrc/attention.hip: attn_qkv_gn_h16_kernel) instead of normalise (2) + GEMM + split + flash + merge
        wkey = ("frag", x.dtype)
        wqkvs = self._wqkv
        if wkey not in wqkvs:
            wqkvs[wkey] = ops.pack_qkv_weight_h16(wqkvs[torch.float32], x.dtype)
        if ops.ATTN_BLOCK_H16_PROJ:
            pkey = ("proj_frag", x.dtype, self.proj_out.weight.data_ptr(), self.proj_out.weight._version)
            if pkey not in wqkvs:
                wqkvs[pkey] = (ops.pack_weight_tp_h16(self.proj_out.weight.detach().reshape(C, C).float().contiguous(), x.dtype),
                               self.proj_out.bias.detach().float().contiguous())
            ob = ops.attn_block_h16(x.reshape(B * n, C), x._gn_partials, self.norm.weight.detach(), self.norm.bias.detach(), self.norm.eps,
                                    wqkvs[wkey], bqkv, C, scale, B=B, proj=wqkvs[pkey])
            out = ob.view(B, H, W, C)
            if hasattr(ob, "_gn_partials"):
                out._gn_partials = ob._gn_partials
            return out
        o = ops.attn_block_h16(x.reshape(B * n, C), x._gn_partials, self.norm.weight.detach(), self.norm.bias.detach(), self.norm.eps,
                               wqkvs[wkey], bqkv, C, scale, B=B)
        return self.proj_out.forward_nhwc(o.view(B, H, W, C), residual=x)
    h = self.norm.forward_nhwc(x, swish=False)
    if ops.attention_small_fits(n, C, B):
        # the 16 x 16 blocks: scores, soft-max and P v of a query tile in ONE launch (transpose + GEMM + soft-max + GEMM otherwise), any batch
        qkv = ops.gemm_nt(h.reshape(B * n, C), wqkv, bias=bqkv)
        o = ops.attention_small(qkv, C, scale, B=B)
        return self.proj_out.forward_nhwc(o.view(B, H, W, C), residual=x)
    if B > 1 and ops.attention_fusable(n, C):
        qkv = ops.gemm_nt(h.reshape(B * n, C), wqkv, bias=bqkv)                    # (B n, 3C): the whole batch in one GEMM
        o = ops.attention_h16(qkv, C, scale, B=B)
        return self.proj_out.forward_nhwc(o.view(B, H, W, C), residual=x)
    if B > 1 and B * n <= BLOCKDIAG_MAX_ROWS and n % 4 == 0:
        qkv = ops.gemm_nt(h.reshape(B * n, C), wqkv, bias=bqkv)                    # block-diagonal chain (forward_nhwc)
        vt = ops.transpose_h16(qkv[:, 2 * C:])
        s = ops.gemm_nt(qkv[:, :C], qkv[:, C:2 * C], out_dtype=torch.float32)
        o = ops.gemm_nt(ops.softmax_rows_h16(s, scale, x.dtype, block=n), vt)
        return self.proj_out.forward_nhwc(o.view(B, H, W, C), residual=x)
    out = torch.empty_like(x)
    for b in range(B):
        qkv = ops.gemm_nt(h[b].reshape(n, C), wqkv, bias=bqkv)                     # (n, 3C) 16-bit
        if ops.attention_fusable(n, C):
            o = ops.attention_h16(qkv, C, scale)                                   # one pass over the keys
        else:
            vt = ops.transpose_h16(qkv[:, 2 * C:])                                 # (C, n)
            s = ops.gemm_nt(qkv[:, :C], qkv[:, C:2 * C], out_dtype=torch.float32)  # (n, n) fp32 scores
            p = ops.softmax_rows_h16(s, scale, x.dtype)                            # (n, n) 16-bit probabilities
            o = ops.gemm_nt(p, vt)                                                 # (n, C)
        ob = ops.gemm_nt(o, wp, bias=bp, residual=x[b].reshape(n, C), out=out[b].reshape(n, C))
        if B == 1 and hasattr(ob, "_gn_partials"):
            out._gn_partials = ob._gn_partials   # statistics of the block output for the next GroupNorm
    return out


AttnBlock._forward_nhwc_h16 = _attn_h16


def _make_attn_list():
    return nn.ModuleList()


class Encoder(_NHWCModule):
    def __init__(self, *, ch, out_ch, ch_mult=(1, 2, 4, 8), num_res_blocks, attn_resolutions, dropout=0.0,
                 resamp_with_conv=True, in_channels, resolution, z_channels, double_z=True, **ignore_kwargs):
        super().__init__()
        self.ch, self.temb_ch = ch, 0
        self.num_resolutions = len(ch_mult)
        self.num_res_blocks = num_res_blocks
        self.resolution = resolution
        self.in_channels = in_channels
        self.conv_in = Conv2d(in_channels, ch, kernel_size=3, stride=1, padding=1)
        # attention placement follows ddconfig.resolution bookkeeping, NOT the real input size
        res = resolution
        widths = [ch * m for m in (1,) + tuple(ch_mult)]
        self.down = nn.ModuleList()
        for lv in range(self.num_resolutions):
            stage = nn.Module()
            stage.block, stage.attn = nn.ModuleList(), _make_attn_list()
            cin, cout = widths[lv], widths[lv + 1]
            for _ in range(num_res_blocks):
                stage.block.append(ResnetBlock(in_channels=cin, out_channels=cout, temb_channels=0, dropout=dropout))
                cin = cout
                if res in attn_resolutions:
                    stage.attn.append(AttnBlock(cin))
            if lv != self.num_resolutions - 1:
                stage.downsample = Downsample(cin, resamp_with_conv)
                res //= 2
            self.down.append(stage)
        top = widths[-1]
        self.mid = nn.Module()
        self.mid.block_1 = ResnetBlock(in_channels=top, out_channels=top, temb_channels=0, dropout=dropout)
        self.mid.attn_1 = AttnBlock(top)
        self.mid.block_2 = ResnetBlock(in_channels=top, out_channels=top, temb_channels=0, dropout=dropout)
        self.norm_out = Normalize(top)
        self.conv_out = Conv2d(top, 2 * z_channels if double_z else z_channels, kernel_size=3, stride=1, padding=1)

    def forward_nhwc(self, x):
        """x: (B,H,W,32) NHWC with the in_channels real channels first, rest zero."""
        h = self.conv_in.forward_nhwc(x)
        for lv, stage in enumerate(self.down):
            for ib, blk in enumerate(stage.block):
                h = blk.forward_nhwc(h)
                if len(stage.attn) > 0:
                    h = stage.attn[ib].forward_nhwc(h)
            if lv != self.num_resolutions - 1:
                h = stage.downsample.forward_nhwc(h)
        h = self.mid.block_2.forward_nhwc(self.mid.attn_1.forward_nhwc(self.mid.block_1.forward_nhwc(h)))
        return _norm_conv(self.norm_out, True, self.conv_out, h)

    def forward(self, x):
        return ops.nhwc_to_nchw(self.forward_nhwc(ops.nchw_to_nhwc(x, c_pad=32)))


class Decoder(_NHWCModule):
    def __init__(self, *, ch, out_ch, ch_mult=(1, 2, 4, 8), num_res_blocks, attn_resolutions, dropout=0.0,
                 resamp_with_conv=True, in_channels, resolution, z_channels, give_pre_end=False, **ignorekwargs):
        super().__init__()
        self.ch, self.temb_ch = ch, 0
        self.num_resolutions = len(ch_mult)
        self.num_res_blocks = num_res_blocks
        self.resolution = resolution
        self.in_channels = in_channels
        self.give_pre_end = give_pre_end
        self.out_ch = out_ch
        width = ch * ch_mult[-1]
        res = resolution // 2 ** (self.num_resolutions - 1)
        self.z_shape = (1, z_channels, res, res)
        print("Working with z of shape {} = {} dimensions.".format(self.z_shape, int(np.prod(self.z_shape))))
        self.conv_in = Conv2d(z_channels, width, kernel_size=3, stride=1, padding=1)
        self.mid = nn.Module()
        self.mid.block_1 = ResnetBlock(in_channels=width, out_channels=width, temb_channels=0, dropout=dropout)
        self.mid.attn_1 = AttnBlock(width)
        self.mid.block_2 = ResnetBlock(in_channels=width, out_channels=width, temb_channels=0, dropout=dropout)
        stages = []
        for lv in reversed(range(self.num_resolutions)):
            stage = nn.Module()
            stage.block, stage.attn = nn.ModuleList(), _make_attn_list()
            cout = ch * ch_mult[lv]
            for _ in range(num_res_blocks + 1):
                stage.block.append(ResnetBlock(in_channels=width, out_channels=cout, temb_channels=0, dropout=dropout))
                width = cout
                if res in attn_resolutions:
                    stage.attn.append(AttnBlock(width))
            if lv != 0:
                stage.upsample = Upsample(width, resamp_with_conv)
                res *= 2
            stages.append(stage)
        self.up = nn.ModuleList(reversed(stages))  # index = resolution level, like the reference
        self.norm_out = Normalize(width)
        self.conv_out = Conv2d(width, out_ch, kernel_size=3, stride=1, padding=1)

    def forward_nhwc(self, z):
        self.last_z_shape = (z.shape[0], z.shape[3], z.shape[1], z.shape[2])
        h = self.conv_in.forward_nhwc(z)
        h = self.mid.block_2.forward_nhwc(self.mid.attn_1.forward_nhwc(self.mid.block_1.forward_nhwc(h)))
        for lv in reversed(range(self.num_resolutions)):
            stage = self.up[lv]
            for ib, blk in enumerate(stage.block):
                h = blk.forward_nhwc(h)
                if len(stage.attn) > 0:
                    h = stage.attn[ib].forward_nhwc(h)
            if lv != 0:
                h = stage.upsample.forward_nhwc(h)
        if self.give_pre_end:
            return h
        return _norm_conv(self.norm_out, True, self.conv_out, h, out_dtype=torch.float32)  # RGB-D leaves in fp32
