"""The reference's training step on the HIP kernels.

What the reference does per batch (sgam/generative_sensing_module/model.py:271-345): forward of the conditional VQGAN,
`VQLPIPSWithDiscriminator.forward(qloss, x_dst, xrec, optimizer_idx=0, global_step, ...)`
(modules/losses/vqperceptual.py:77-110), `opt_ae.zero_grad(); aeloss.backward(); opt_ae.step()`, then the same with
`optimizer_idx=1` for the PatchGAN discriminator.  Built here:

  `AutoencoderTrainer`   loss = mean|x_dst - xrec| + codebook_weight * qloss   (what the reference's loss reduces to before
                          `disc_start`), Adam(lr, betas=(0.5, 0.9)) over the phase's parameter set (model.py:414-428:
                          `conditional_generation`: encoder + conv_in; `codebook`: the whole autoencoder incl. the embedding)
  `VQGANTrainer`          the whole step: + g_loss = -mean D(xrec) weighted by the adaptive `d_weight` = ||d nll / d W_last|| /
                          (||d g_loss / d W_last|| + 1e-4) * disc_weight and by `disc_factor` (0 before `disc_start`), then the
                          discriminator update with the hinge loss on D(x_dst), D(xrec.detach()) and its own Adam; the three
                          discriminator forwards per step update its BatchNorm running statistics like the reference's

  LPIPS (`_Lpips`)        rec_loss = |x - xrec| + perceptual_weight * LPIPS(x[:, :3], xrec[:, :3]) (vqperceptual.py:79-84): the
                          ScalingLayer, the frozen VGG16 trunk on both images, per level normalise / squared difference / `lin` /
                          spatial mean, and the backward pass to the reconstruction.  The trunk's ImageNet checkpoint is
                          torchvision's and cannot be fetched here: tests run on synthetic trunk weights (the `lin` weights are the
                          reference's shipped ones), so what is pinned is the computation, not the metric's pretrained values.
  `OnlineCodebookRefresh` the online k-means refresh of dead codewords (model.py:274-295, 313-323): host logic with scipy's
                          kmeans2 in the reference and here.

Arithmetic: every product (forward convolutions, data / weight gradients, attention) runs on the MFMA GEMM of csrc/conv_gemm.hip
in its fp32-in mode (`ops.set_f32_mode("mfma")` for the duration of a step: gradients sit far below fp16's normal range, so the
exact hi / lo fp16 split of the inference path does not apply); the index, reduction and element-wise kernels are
csrc/train.hip.  torch is used for memory, views, zero-padded copies and `torch.distributed` only, never for arithmetic
(the few hundred per-workgroup partial sums behind a logged scalar or a gradient norm are added on the host, where the
reference calls `.item()`).  Activations NHWC fp32, one tape entry per layer.
"""
import contextlib
import ctypes

import torch

from . import _lib, ops
from ._lib import ConvDesc
from .ops import _p, _stream, check

ADAM_BETAS = (0.5, 0.9)          # model.py:423-432
DGRAD_AS_CONV = True             # 3x3 / s1 / p1 data gradients as a convolution with the flipped filter (False: GEMM + col2im gather)
ADAM_EPS = 1e-8


def _round_up(v, m):
    return (v + m - 1) // m * m


def _transpose2d(a):
    """(R, C) dense fp32 -> (C, R) dense (the layout-hop kernel of the forward path)"""
    R, C = a.shape
    return ops.nhwc_to_nchw(a.reshape(1, R, 1, C)).reshape(C, R)


def _pad_rows(a2d, rows):
    """(R, C) -> (rows, C) with a zero tail (a copy; no arithmetic)"""
    if a2d.shape[0] == rows:
        return a2d
    out = torch.zeros((rows, a2d.shape[1]), device=a2d.device, dtype=a2d.dtype)
    out[:a2d.shape[0]] = a2d
    return out


def _pad_channels(x, c):
    """(..., C) -> (..., c) dense with zero channels appended (a copy; no arithmetic)"""
    if x.shape[-1] == c:
        return x if x.is_contiguous() else x.contiguous()
    out = torch.zeros(x.shape[:-1] + (c,), device=x.device, dtype=x.dtype)
    out[..., :x.shape[-1]] = x
    return out


def _host_sum(partial):
    """fold of a few hundred per-workgroup partial sums (logging values and the two norms of the adaptive weight)"""
    return float(partial.cpu().numpy().sum())


def _axpby(a, b=None, alpha=1.0, beta=1.0):
    out = torch.empty_like(a)
    check(_lib.load().sgam_axpby_f32(_p(a), _p(b), _p(out), a.numel(), float(alpha), float(beta), _stream()), "sgam_axpby_f32")
    return out


def _colsum(a2d):
    M, N = a2d.shape
    lib = _lib.load()
    nb = lib.sgam_colsum_workspace_bytes(M, N)
    ws = torch.empty((nb,), device=a2d.device, dtype=torch.uint8)
    out = torch.empty((N,), device=a2d.device, dtype=torch.float32)
    check(lib.sgam_colsum_f32(_p(a2d), a2d.stride(0), _p(out), M, N, _p(ws), nb, _stream()), "sgam_colsum_f32")
    return out


class _Conv:
    """one Conv2d of the model: forward through ops.conv2d_nhwc, backward as two GEMMs + the index kernels of train.hip"""

    def __init__(self, conv, grads, upsample2x=False, pad=None, need_wgrad=True):
        self.conv, self.grads, self.ups, self.need_wgrad = conv, grads, upsample2x, need_wgrad
        kh, kw = conv.kernel_size
        self.kh, self.kw, self.stride = kh, kw, conv.stride[0]
        self.pad = pad if pad is not None else (conv.padding[0], conv.padding[1], conv.padding[0], conv.padding[1])   # t, l, b, r
        self.cin, self.cout = conv.in_channels, conv.out_channels
        self.cin_pad = _round_up(self.cin, 32)
        self.cout_k = _round_up(self.cout, 32)            # the K of the data-gradient GEMM (dy is that wide)

    def fwd(self, x, residual=None):
        conv = self.conv
        wp, b = conv._packed(torch.float32)
        assert x.shape[3] == self.cin_pad, (x.shape, self.cin_pad)
        self.x = x
        y = ops.conv2d_nhwc(x, wp, b, cout=self.cout, kh=self.kh, kw=self.kw, stride=self.stride, pad_t=self.pad[0],
                            pad_l=self.pad[1], pad_b=self.pad[2], pad_r=self.pad[3], upsample2x=self.ups, residual=residual,
                            cin=self.cin_pad)
        B, Hi, Wi, _ = x.shape
        self.desc = ConvDesc(B=B, Hi=Hi, Wi=Wi, Cin=self.cin_pad, Ho=y.shape[1], Wo=y.shape[2], N=self.cout, KH=self.kh, KW=self.kw,
                             stride=self.stride, pad_t=self.pad[0], pad_l=self.pad[1], upsample2x=int(self.ups), lda=self.cin_pad,
                             ldb=0, ldc=self.cout, ldr=0, n_valid=self.cout, bias_per_row=0)
        return y

    def bwd(self, dy, need_dx=True):
        """dy (B,Ho,Wo,cout_k) dense (columns >= cout are zero padding) -> dx (B,Hi,Wi,cin_pad) or None"""
        lib = _lib.load()
        d, x = self.desc, self.x
        M = d.B * d.Ho * d.Wo
        K = self.kh * self.kw * self.cin_pad
        dy2 = dy.reshape(M, dy.shape[-1])
        assert dy2.shape[1] == self.cout_k, (dy2.shape, self.cout_k)
        pointwise = self.kh == 1 and self.kw == 1 and self.stride == 1 and not self.ups
        if self.need_wgrad:
            Mp = _round_up(M, 32)                                          # the weight-gradient GEMM contracts over M: zero tail
            dyT = _transpose2d(_pad_rows(dy2, Mp))                         # [cout_k][Mp]
            if pointwise:
                colT = _transpose2d(_pad_rows(x.reshape(M, self.cin_pad), Mp))
            else:
                colT = (torch.empty if Mp == M else torch.zeros)((K, Mp), device=x.device, dtype=torch.float32)
                check(lib.sgam_im2col_t_f32(ctypes.byref(d), _p(x), _p(colT), self.cin_pad, Mp, _stream()), "sgam_im2col_t_f32")
            gwp = ops.gemm_nt(dyT, colT)                                   # [cout_k][K]
            gw = torch.empty_like(self.conv.weight)
            check(lib.sgam_unpack_conv_weight_grad_f32(_p(gwp), gwp.stride(0), _p(gw), self.cout, self.cin, self.kh, self.kw,
                                                       self.cin_pad, _stream()), "sgam_unpack_conv_weight_grad_f32")
            self.grads[self.conv.weight] = gw
            if self.conv.bias is not None:
                self.grads[self.conv.bias] = _colsum(dy2)[:self.cout].contiguous() if self.cout_k != self.cout else _colsum(dy2)
        if not need_dx:
            return None
        # W^T [K][cout_k]: the packed rows (padded to cout_k) transposed.  A layer that is frozen in this phase (need_wgrad False: e.g.
        # the whole decoder while `conditional_generation` trains the encoder) keeps the copy across steps (two launches fewer per
        # layer and step); trained weights are re-packed every step (Adam writes through the raw pointer: no version bump to key on)
        w = self.conv.weight
        wkey = (w.data_ptr(), w._version)
        if self.kh == 3 and self.kw == 3 and self.stride == 1 and not self.ups and self.pad == (1, 1, 1, 1) and self.cin == self.cin_pad \
                and DGRAD_AS_CONV:
            # the data gradient of a 3x3 / stride 1 / pad 1 convolution IS such a convolution — of dy with the spatially flipped,
            # channel-transposed filter: dx[i] = sum_k dy[i + k - 1] . W[:, :, 2 - k]^T — so it runs on the forward implicit-GEMM
            # kernel: no [M][9 Cin] column matrix (302 MB for a 128-channel layer at 256^2) is written and gathered back
            if not self.need_wgrad and getattr(self, "_wd_key", None) == wkey:
                wd = self._wd
            else:
                wd = ops.pack_conv_weight(w.detach().flip(2, 3).transpose(0, 1).contiguous(), cin_pad=self.cout_k, dtype=torch.float32)
                if not self.need_wgrad:
                    self._wd, self._wd_key = wd, wkey
            return ops.conv2d_nhwc(dy, wd, None, cout=self.cin, kh=3, kw=3, stride=1, pad_t=1, pad_l=1, cin=self.cout_k)
        if not self.need_wgrad and getattr(self, "_wT_key", None) == wkey:
            wT = self._wT
        else:
            wT = _transpose2d(ops.pack_conv_weight(w, cout_pad=self.cout_k, cin_pad=self.cin_pad, dtype=torch.float32))
            if not self.need_wgrad:
                self._wT, self._wT_key = wT, wkey
        dcol = ops.gemm_nt(dy2, wT)                                        # [M][K]
        if pointwise:
            return dcol.reshape(x.shape)
        dx = torch.empty_like(x)
        check(lib.sgam_col2im_gather_f32(ctypes.byref(d), _p(dcol), _p(dx), self.cin_pad, _stream()), "sgam_col2im_gather_f32")
        return dx


class _Norm:
    """GroupNorm(32, eps=1e-6)(+swish) (Normalize / nonlinearity, diffusionmodules/model.py:30-40)"""

    def __init__(self, norm, swish, grads, need_pgrad=True):
        self.norm, self.swish, self.grads, self.need_pgrad = norm, swish, grads, need_pgrad

    def fwd(self, x):
        self.x = x
        self.mr = ops.groupnorm_meanrstd(x, self.norm.eps)
        return ops.groupnorm_nhwc(x, self.norm.weight.detach(), self.norm.bias.detach(), self.swish, groups=self.norm.num_groups,
                                  eps=self.norm.eps)

    def bwd(self, dy):
        x = self.x
        B, H, W, C = x.shape
        G = self.norm.num_groups
        dx = torch.empty_like(x)
        dg = torch.empty((B, C), device=x.device, dtype=torch.float32)
        db = torch.empty((B, C), device=x.device, dtype=torch.float32)
        gm = torch.empty((B, G, 2), device=x.device, dtype=torch.float32)
        lib = _lib.load()
        nb = lib.sgam_groupnorm_bwd_workspace_bytes(B, H * W, C)
        ws = torch.empty((nb,), device=x.device, dtype=torch.uint8)
        check(lib.sgam_groupnorm_bwd_nhwc_f32(_p(x), _p(dy), _p(self.mr), _p(ops._f32c(self.norm.weight.detach())),
                                              _p(ops._f32c(self.norm.bias.detach())), int(self.swish), _p(dx), _p(dg), _p(db), _p(gm), B,
                                              H * W, C, G, _p(ws), nb, _stream()), "sgam_groupnorm_bwd_nhwc_f32")
        if self.need_pgrad:
            self.grads[self.norm.weight] = _colsum(dg) if B > 1 else dg.reshape(C)
            self.grads[self.norm.bias] = _colsum(db) if B > 1 else db.reshape(C)
        return dx


class _ResBlock:
    """ResnetBlock (model.py:78-137, temb = None, dropout 0): x + conv2(swish(norm2(conv1(swish(norm1(x))))))"""

    def __init__(self, blk, grads, train):
        self.n1, self.c1 = _Norm(blk.norm1, True, grads, train), _Conv(blk.conv1, grads, need_wgrad=train)
        self.n2, self.c2 = _Norm(blk.norm2, True, grads, train), _Conv(blk.conv2, grads, need_wgrad=train)
        self.sc = None
        if blk.in_channels != blk.out_channels:
            self.sc = _Conv(blk.conv_shortcut if blk.use_conv_shortcut else blk.nin_shortcut, grads, need_wgrad=train)

    def fwd(self, x):
        h = self.c1.fwd(self.n1.fwd(x))
        s = self.sc.fwd(x) if self.sc is not None else x
        return self.c2.fwd(self.n2.fwd(h), residual=s)

    def bwd(self, dy):
        dh = self.n1.bwd(self.c1.bwd(self.n2.bwd(self.c2.bwd(dy))))
        ds = self.sc.bwd(dy) if self.sc is not None else dy
        return _axpby(dh, ds)


class _Attn:
    """AttnBlock (model.py:140-192): x + proj_out(softmax(q k^T c^-1/2) v) on norm(x), one image at a time, one panel of query
    rows at a time"""

    def __init__(self, att, grads, train):
        self.att, self.grads = att, grads
        self.norm = _Norm(att.norm, False, grads, train)
        self.q, self.k, self.v = (_Conv(m, grads, need_wgrad=train) for m in (att.q, att.k, att.v))
        self.proj = _Conv(att.proj_out, grads, need_wgrad=train)

    CHUNK_BYTES = 64 << 20      # budget of one (rows x n) fp32 score panel: 4096 x 4096 in one piece, 1024 rows at n = 16384

    def _rows(self, n):
        r = min(n, (self.CHUNK_BYTES // (4 * n)) // 32 * 32)
        while r >= 32 and n % r:
            r -= 32
        return r if r >= 32 else n          # n < 32, or no multiple of 32 divides n (a 70 x 70 map): one panel of all rows

    def fwd(self, x):
        B, H, W, C = x.shape
        n = H * W
        h = self.norm.fwd(x)
        q, k, v = self.q.fwd(h), self.k.fwd(h), self.v.fwd(h)
        self.scale = float(C) ** -0.5
        # the tape keeps q, k, v only (3 n C floats per image): the n x n probabilities are RECOMPUTED in the backward pass, one
        # panel of query rows at a time (flash-style at the GEMM level) — at n = 4096 that is 64 MB per image and block that no
        # longer sit on the tape across the whole step, at n = 16384 (1 GiB per image) it is what makes the step fit at all
        self.saved = (q, k, v)
        o = torch.empty_like(q)
        R = self._rows(n)
        for b in range(B):
            qb, kb, vb = (t[b].reshape(n, C) for t in (q, k, v))
            vT = _transpose2d(vb)
            ob = o[b].reshape(n, C)
            for r0 in range(0, n, R):
                p = ops.softmax_rows_(ops.gemm_nt(qb[r0:r0 + R], kb), self.scale)          # [R][n] = softmax over the keys
                ops.gemm_nt(p, vT, out=ob[r0:r0 + R])
        return self.proj.fwd(o, residual=x)

    def bwd(self, dy):
        lib = _lib.load()
        do = self.proj.bwd(dy)
        B, H, W, C = do.shape
        n = H * W
        q, k, v = self.saved
        dq, dk, dv = (torch.empty_like(do) for _ in range(3))
        R = self._rows(n)
        for b in range(B):
            qb, kb, vb = (t[b].reshape(n, C) for t in (q, k, v))
            dob, dqb, dkb, dvb = (t[b].reshape(n, C) for t in (do, dq, dk, dv))
            kT = _transpose2d(kb)
            for r0 in range(0, n, R):
                first = r0 == 0
                qc, doc = qb[r0:r0 + R], dob[r0:r0 + R]
                p = ops.softmax_rows_(ops.gemm_nt(qc, kb), self.scale)               # recomputed probabilities of these queries
                dp = ops.gemm_nt(doc, vb)                                            # dO v^T
                ops.gemm_nt(_transpose2d(p), _transpose2d(doc), out=dvb, residual=None if first else dvb)      # dV += P^T dO
                ds = torch.empty_like(p)
                check(lib.sgam_softmax_bwd_rows_f32(_p(p), _p(dp), _p(ds), R, n, p.stride(0), self.scale, _stream()),
                      "sgam_softmax_bwd_rows_f32")
                ops.gemm_nt(ds, kT, out=dqb[r0:r0 + R])                              # dQ = dS k
                ops.gemm_nt(_transpose2d(ds), _transpose2d(qc), out=dkb, residual=None if first else dkb)      # dK += dS^T q
        dh = _axpby(_axpby(self.q.bwd(dq), self.k.bwd(dk)), self.v.bwd(dv))
        return _axpby(self.norm.bwd(dh), dy)


class _Seq:
    def __init__(self, layers):
        self.layers = layers

    def fwd(self, x):
        for l in self.layers:
            x = l.fwd(x)
        return x

    def bwd(self, dy):
        for l in reversed(self.layers):
            dy = l.bwd(dy)
        return dy


def _encoder_layers(enc, grads, train):
    ls = [_Conv(enc.conv_in, grads, need_wgrad=train)]
    for lv, stage in enumerate(enc.down):
        for ib, blk in enumerate(stage.block):
            ls.append(_ResBlock(blk, grads, train))
            if len(stage.attn) > 0:
                ls.append(_Attn(stage.attn[ib], grads, train))
        if lv != enc.num_resolutions - 1:
            ls.append(_Conv(stage.downsample.conv, grads, pad=(0, 0, 1, 1), need_wgrad=train))
    ls += [_ResBlock(enc.mid.block_1, grads, train), _Attn(enc.mid.attn_1, grads, train), _ResBlock(enc.mid.block_2, grads, train),
           _Norm(enc.norm_out, True, grads, train), _Conv(enc.conv_out, grads, need_wgrad=train)]
    return ls


def _decoder_layers(dec, grads, train):
    ls = [_Conv(dec.conv_in, grads, need_wgrad=train), _ResBlock(dec.mid.block_1, grads, train), _Attn(dec.mid.attn_1, grads, train),
          _ResBlock(dec.mid.block_2, grads, train)]
    for lv in reversed(range(dec.num_resolutions)):
        stage = dec.up[lv]
        for ib, blk in enumerate(stage.block):
            ls.append(_ResBlock(blk, grads, train))
            if len(stage.attn) > 0:
                ls.append(_Attn(stage.attn[ib], grads, train))
        if lv != 0:
            ls.append(_Conv(stage.upsample.conv, grads, upsample2x=True, need_wgrad=train))
    ls += [_Norm(dec.norm_out, True, grads, train), _Conv(dec.conv_out, grads, need_wgrad=train)]
    return ls


class AdamHandle:
    """What `VQModel.configure_optimizers` returns (model.py:405-432): the parameter set, hyper-parameters and state of one of
    the step's two Adam optimisers.  The update itself is `sgam_adam_step_f32` (csrc/train.hip), applied by the trainer inside
    `training_step` — the reference, too, steps its optimisers by hand there (`automatic_optimization = False`)."""

    def __init__(self, params, lr, state):
        self.param_groups = [{"params": list(params), "lr": lr, "betas": ADAM_BETAS, "eps": ADAM_EPS}]
        self.state = state           # parameter -> (exp_avg, exp_avg_sq), filled on the first step

    def zero_grad(self, set_to_none=True):
        """gradients live in the trainer's tape and are rebuilt every step: nothing to clear"""

    def step(self):
        raise ops.SgamHipError("the HIP trainer applies Adam inside VQModel.training_step (training.VQGANTrainer.step)")


class AutoencoderTrainer:
    """`loss, log = trainer.step(x, x_dst, extrapolation_mask)`: one autoencoder update of VQModel.training_step (see the module
    docstring for what is and is not built).  `x` is what `get_x` / `get_input` hands to the model (B,4,H,W), `x_dst` the
    reconstruction target; `phase` selects the parameter set like `configure_optimizers`."""

    def __init__(self, model, phase=None, lr=None, codebook_weight=1.0, process_group=None):
        self.model = model
        self.phase = phase or getattr(model, "phase", "conditional_generation")
        if self.phase not in ("conditional_generation", "codebook"):
            raise NotImplementedError(self.phase)
        self.lr = float(lr if lr is not None else getattr(model, "learning_rate", 4.5e-6))
        self.codebook_weight = float(codebook_weight)
        self.pg = process_group
        self.global_step = 0
        self.state = {}                 # parameter -> (exp_avg, exp_avg_sq)
        self.grads = {}
        full = self.phase == "codebook"
        self.enc = _Seq(_encoder_layers(model.encoder, self.grads, True))
        self.quant_conv = _Conv(model.quant_conv, self.grads, need_wgrad=full)
        self.post_quant_conv = _Conv(model.post_quant_conv, self.grads, need_wgrad=full)
        self.dec = _Seq(_decoder_layers(model.decoder, self.grads, full))
        self.head = _Conv(model.conv_in, self.grads) if model.use_extrapolation_mask else None
        self.refresh = None
        kcfg = getattr(model, "online_kmeans_config", None)
        if self.phase == "codebook" and kcfg and kcfg.get("do_online_kmeans_clustering"):
            rank = torch.distributed.get_rank() if (torch.distributed.is_available() and torch.distributed.is_initialized()) else 0
            self.refresh = OnlineCodebookRefresh(model, kcfg, rank)
        self.synced_tensors = self._broadcast_module(model)          # world > 1: start every rank from rank 0's weights

    # ---- what DistributedDataParallel does at construction: every rank starts from rank 0's parameters and buffers
    def _world(self):
        import torch.distributed as dist
        return dist.get_world_size(self.pg) if (dist.is_available() and dist.is_initialized()) else 1

    def _src_rank(self):
        import torch.distributed as dist
        return dist.get_global_rank(self.pg, 0) if self.pg is not None else 0

    def _broadcast_module(self, module):
        """rank 0's parameters AND buffers (BatchNorm running statistics, num_batches_tracked) to every rank; returns the
        number of tensors sent.  Differences from DDP that remain: buffers are synchronised here once, not before every
        forward (`broadcast_buffers`) — the PatchGAN's running statistics then evolve per rank from per-rank batches, they
        are not used by the training-mode forward — and gradients are averaged through one flat bucket after the whole
        backward instead of bucket by bucket during it."""
        import torch.distributed as dist
        from .distributed import collectives_active
        if not collectives_active(self.pg):
            return 0
        n = 0
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t.data, src=self._src_rank(), group=self.pg)
            n += 1
        _invalidate_packs(module)
        return n

    # ---- the parameter set of the phase, in the order of configure_optimizers (model.py:414-428)
    def parameters(self):
        m = self.model
        ps = list(m.encoder.parameters())
        if self.phase == "codebook":
            ps += list(m.decoder.parameters()) + list(m.quantize.parameters()) + list(m.quant_conv.parameters()) + \
                list(m.post_quant_conv.parameters())
        if m.use_extrapolation_mask:
            ps += list(m.conv_in.parameters())
        return ps

    def _input_nhwc(self, x, mask):
        B, C, H, W = x.shape
        if self.head is None:
            return ops.nchw_to_nhwc(x, c_pad=32)
        # cat(x, mask) -> 1x1 conv (model.py:107-113): laid out as a 32-channel NHWC tensor with 5 real channels
        if mask is None:
            mask = torch.zeros((B, 1, H, W), device=x.device)
        m = mask.reshape(B, 1, H, W).to(torch.float32)
        x5 = ops.nchw_to_nhwc(torch.cat([x.to(torch.float32), m], 1), c_pad=32)
        h = self.head.fwd(x5)                                        # (B,H,W,4)
        out = torch.zeros((B, H, W, 32), device=x.device, dtype=torch.float32)
        out[..., :4] = h
        return out

    def _forward(self, x, x_dst, extrapolation_mask):
        """forward with tape + the reconstruction loss and its gradient (call inside _mfma_mode())"""
        m, lib = self.model, _lib.load()
        if hasattr(m, "use_vq") and not m.use_vq():
            # before vq_step_threshold the reference's forward skips the quantiser (model.py:148-150): a different graph
            raise NotImplementedError("training step before vq_step_threshold (un-quantised forward) is not built")
        self.grads.clear()
        if self.refresh is not None:
            self.refresh.before_step(self.global_step)
            if self._world() > 1 and self.refresh.may_fire(self.global_step):
                # the refresh rewrites codebook rows on rank 0 only (model.py:276, as in the reference — whose other ranks then
                # keep the stale rows): every rank can tell from the step counter WHEN that may have happened, so the codebook
                # is re-broadcast on exactly those steps
                import torch.distributed as dist
                dist.broadcast(m.quantize.embedding.weight.data, src=self._src_rank(), group=self.pg)
                _invalidate_packs(m)
        xin = self._input_nhwc(x, extrapolation_mask)
        z = self.quant_conv.fwd(self.enc.fwd(xin))                               # (B,h,w,D)
        zq_st, idx, _ = m.quantize.quantize_nhwc(z)                               # straight-through value, indices
        e = ops.vq_gather(m.quantize._codebook()[0], idx).view(z.shape)           # the codebook rows themselves
        qloss = float(m.quantize.commit_loss_nhwc(z, idx))
        if self.refresh is not None and self.refresh._started(self.global_step) and self.refresh.rank == 0:
            self.refresh.after_forward(self.global_step, idx.cpu().numpy(), ops.nhwc_to_nchw(z[:1])[0].cpu().numpy())
        rec = self.dec.fwd(self.post_quant_conv.fwd(zq_st))                       # (B,H,W,out_ch)
        C = rec.shape[3]
        rows = rec.numel() // C
        tgt = ops.nchw_to_nhwc(x_dst)                                             # (B,H,W,C)
        ldg = _round_up(C, 32)
        drec = torch.empty(rec.shape[:3] + (ldg,), device=rec.device, dtype=torch.float32)
        part = torch.empty(((rows * ldg + 255) // 256,), device=rec.device, dtype=torch.float64)
        check(lib.sgam_l1_loss_grad_f32(_p(rec), _p(tgt), _p(drec), _p(part), rows, C, C, ldg, 1.0 / (rows * C), _stream()),
              "sgam_l1_loss_grad_f32")
        # logging values are folded on the host like the reference's `.item()`s (the backward pass does not depend on them)
        return {"z": z, "e": e, "idx": idx, "rec": rec, "drec": drec, "nll": _host_sum(part) / (rows * C), "qloss": qloss}

    def _backward(self, fw, drec):
        """drec (B,H,W,32) = dL/d(reconstruction) -> self.grads for every parameter of the phase"""
        m, lib = self.model, _lib.load()
        z, e, idx = fw["z"], fw["e"], fw["idx"]
        B, h, w, D = z.shape
        dzq = self.post_quant_conv.bwd(self.dec.bwd(drec))
        two_c = 2.0 * self.codebook_weight / z.numel()
        dz = torch.empty_like(z)
        check(lib.sgam_vq_bwd_f32(_p(dzq), _p(z), _p(e), _p(dz), z.numel(), two_c, _stream()), "sgam_vq_bwd_f32")
        if self.phase == "codebook":
            emb = m.quantize.embedding.weight
            ge = torch.empty_like(emb)
            check(lib.sgam_vq_codebook_grad_f32(_p(idx.reshape(-1)), _p(z), _p(e), _p(ge), B * h * w, emb.shape[0], D,
                                                two_c * m.quantize.beta, _stream()), "sgam_vq_codebook_grad_f32")
            self.grads[emb] = ge
        dxin = self.enc.bwd(self.quant_conv.bwd(dz))
        if self.head is not None:
            self.head.bwd(_pad_channels(dxin[..., :4], 32), need_dx=False)

    def forward_backward(self, x, x_dst, extrapolation_mask=None):
        """-> dict of loss terms / outputs; fills self.grads for every parameter of the phase"""
        with _mfma_mode():
            fw = self._forward(x, x_dst, extrapolation_mask)
            self._backward(fw, fw["drec"])
        return {"nll_loss": fw["nll"], "quant_loss": fw["qloss"], "loss": fw["nll"] + self.codebook_weight * fw["qloss"],
                "rec": fw["rec"], "indices": fw["idx"]}

    def allreduce_grads(self):
        """the gradient half of what DDP does for the reference's LightningModule: average the gradients over the ranks — one
        flat bucket, one RCCL all-reduce (111 MB for the encoder set, 276 MB for the whole autoencoder at fp32).  The other
        half — identical starting weights / buffers on every rank — is `_broadcast_module` at construction."""
        self.last_allreduce_bytes = self._allreduce(self.grads, self.parameters())
        return self.last_allreduce_bytes

    def _allreduce(self, grads, params):
        import torch.distributed as dist
        from .distributed import collectives_active
        if not collectives_active(self.pg):
            return 0
        ps = [p for p in params if p in grads]
        if not ps:
            return 0
        flat = torch.cat([grads[p].reshape(-1) for p in ps])
        dist.all_reduce(flat, group=self.pg)
        ws, o = dist.get_world_size(self.pg), 0
        for p in ps:
            n = p.numel()
            grads[p] = _axpby(flat[o:o + n].reshape(p.shape).contiguous(), None, 1.0 / ws) if flat.is_cuda else \
                (flat[o:o + n] / ws).reshape(p.shape)
            o += n
        return flat.numel() * 4

    ADAM_CHUNK = 4096          # elements per workgroup of sgam_adam_multi_step_f32 (csrc/train.hip)

    def _adam(self, params, grads, state):
        """opt.step(): Adam on every tensor of `params` that has a gradient — ONE launch for the whole set
        (sgam_adam_multi_step_f32: device tables of pointers; only the gradient pointers change from step to step and travel
        through a pinned staging buffer), instead of one launch per tensor."""
        lib = _lib.load()
        ps = [p for p in params if grads.get(p) is not None]
        if not ps:
            return
        for p in ps:
            if p not in state:
                state[p] = (torch.zeros_like(p.data), torch.zeros_like(p.data))
        dev = ps[0].device
        tabs = self.__dict__.setdefault("_adam_tabs", {})
        # keyed on the storage addresses themselves: a parameter re-homed by `.to()` / an assign-style load, or a restored optimiser
        # state, changes data_ptr() under an unchanged id() and the table must be rebuilt, not written through
        key = tuple((p.data.data_ptr(), state[p][0].data_ptr(), state[p][1].data_ptr(), p.numel()) for p in ps)
        if len(tabs) > 8:
            tabs.clear()
        tab = tabs.get(key)
        if tab is None:
            i64 = lambda v: torch.tensor(v, dtype=torch.int64, device=dev)  # noqa: E731
            bt, bo = [], []
            for t, p in enumerate(ps):
                for o in range(0, p.numel(), self.ADAM_CHUNK):
                    bt.append(t)
                    bo.append(o)
            tab = tabs[key] = {"p": i64([p.data.data_ptr() for p in ps]), "m": i64([state[p][0].data_ptr() for p in ps]),
                               "v": i64([state[p][1].data_ptr() for p in ps]), "n": i64([p.numel() for p in ps]),
                               "bt": torch.tensor(bt, dtype=torch.int32, device=dev), "bo": i64(bo), "blocks": len(bt),
                               "g": torch.empty((len(ps),), dtype=torch.int64, device=dev),
                               "g_host": torch.empty((len(ps),), dtype=torch.int64).pin_memory(), "done": None}
        gs = [ops._f32c(grads[p]) for p in ps]          # (kept alive until the launch is enqueued)
        if tab["done"] is not None:
            tab["done"].synchronize()                  # the previous step's upload has left the staging buffer
        tab["g_host"].copy_(torch.tensor([g.data_ptr() for g in gs], dtype=torch.int64))
        tab["g"].copy_(tab["g_host"], non_blocking=True)
        tab["done"] = torch.cuda.Event()
        tab["done"].record()
        check(lib.sgam_adam_multi_step_f32(_p(tab["p"]), _p(tab["g"]), _p(tab["m"]), _p(tab["v"]), _p(tab["n"]), _p(tab["bt"]),
                                           _p(tab["bo"]), tab["blocks"], self.lr, ADAM_BETAS[0], ADAM_BETAS[1], ADAM_EPS,
                                           self.global_step, _stream()), "sgam_adam_multi_step_f32")

    def adam_step(self):
        self.global_step += 1
        self._adam(self.parameters(), self.grads, self.state)
        _invalidate_packs(self.model)

    def step(self, x, x_dst, extrapolation_mask=None):
        out = self.forward_backward(x, x_dst, extrapolation_mask)
        self.allreduce_grads()
        self.adam_step()
        log = {"train/total_loss": out["loss"], "train/quant_loss": out["quant_loss"], "train/rec_loss": out["nll_loss"],
               "train/nll_loss": out["nll_loss"]}
        return out["loss"], log


class OnlineCodebookRefresh:
    """The online k-means codebook refresh of VQModel.training_step (model.py:274-295 before the forward, :313-323 after it;
    phase `codebook`, rank 0): every codeword carries a countdown that is reset whenever the word is used by the first image
    of a batch; when more than `inactive_threshold` of the words have run out, enough pre-quantisation feature maps are
    buffered and the step is a multiple of `frequency`, the dead words are replaced by the k-means centres of the buffered
    features (`scipy.cluster.vq.kmeans2(..., minit='points')`, the reference's own host call).  Host logic throughout, as in
    the reference; the codebook rows are written through `VectorQuantizer2.update_codebook`."""

    def __init__(self, model, config, rank=0):
        self.model, self.cfg, self.rank = model, dict(config), rank
        self.enabled = bool(self.cfg.get("do_online_kmeans_clustering", False))
        n = model.quantize.embedding.weight.shape[0]
        self.countdown = {i: self.cfg.get("online_kmeans_word_timeout", 10) for i in range(n)}            # train_codebook_map
        self.features = []                                                                               # train_sampled_feature_maps

    def _started(self, global_step):
        return self.enabled and global_step >= self.cfg.get("start_global_step", 0)

    def may_fire(self, global_step):
        """the part of before_step's condition that every rank can evaluate (the countdowns and the buffer live on rank 0)"""
        return self._started(global_step) and global_step % self.cfg["frequency"] == 0

    def before_step(self, global_step):
        """model.py:274-295; returns the number of codewords replaced (0 = none)"""
        if not self._started(global_step) or self.rank != 0:
            return 0
        import numpy as np
        from scipy.cluster.vq import kmeans2
        dead = [k for k, v in self.countdown.items() if v <= 0]
        if not (len(dead) / len(self.countdown) > self.cfg["inactive_threshold"]
                and len(self.features) >= self.cfg["train_feature_buffer_size"] and global_step % self.cfg["frequency"] == 0):
            return 0
        f = np.stack(self.features).transpose(0, 2, 3, 1)
        centres = kmeans2(f.reshape(-1, f.shape[-1]), len(dead), minit="points")[0]
        self.model.quantize.update_codebook(centres.astype(np.float32), dead)
        _invalidate_packs(self.model)
        for k in dead:
            self.countdown[k] = self.cfg["online_kmeans_word_timeout"]
        return len(dead)

    def after_forward(self, global_step, indices, pre_quant_nchw0):
        """model.py:313-323: `indices` = codebook indices of the batch (first image is what the reference looks at),
        `pre_quant_nchw0` = the first image's pre-quantisation features (D,h,w) as a host array"""
        if not self._started(global_step) or self.rank != 0:
            return
        import numpy as np
        for v in np.unique(np.asarray(indices[0]).reshape(-1)):
            self.countdown[int(v)] = self.cfg["online_kmeans_word_timeout"]
        if len(self.features) > self.cfg["train_feature_buffer_size"]:
            self.features = self.features[-self.cfg["train_feature_buffer_size"]:]
        self.features.append(np.asarray(pre_quant_nchw0))
        for k in self.countdown:
            self.countdown[k] -= 1


class _MaxPool:
    def fwd(self, x):
        self.x = x
        B, H, W, C = x.shape
        y = torch.empty((B, H // 2, W // 2, C), device=x.device, dtype=torch.float32)
        check(_lib.load().sgam_maxpool2x2_f32(_p(x), _p(y), B, H, W, C, _stream()), "sgam_maxpool2x2_f32")
        return y

    def bwd(self, dy):
        x = self.x
        B, H, W, C = x.shape
        dx = torch.empty_like(x)
        check(_lib.load().sgam_maxpool2x2_bwd_f32(_p(x), _p(dy), _p(dx), B, H, W, C, _stream()), "sgam_maxpool2x2_bwd_f32")
        return dx


class _Lpips:
    """LPIPS(input, target) of the reference (modules/losses/lpips.py:41-55) and its gradient w.r.t. `input`: ScalingLayer ->
    VGG16 trunk (frozen; 13 convs + ReLU, 4 max-pools) on both images -> per level: unit-normalise over channels, squared
    difference, 1x1 `lin`, spatial mean -> sum over the five levels.  Eval mode (the `lin` dropouts are identities)."""

    def __init__(self, lp):
        self.lp = lp
        self.shift = (ctypes.c_float * 4)(*[float(v) for v in lp.scaling_layer.shift.reshape(-1)], 0.0)
        self.zero = (ctypes.c_float * 4)(0.0, 0.0, 0.0, 0.0)
        self.inv_scale = (ctypes.c_float * 4)(*[1.0 / float(v) for v in lp.scaling_layer.scale.reshape(-1)], 0.0)

    def _trunk(self):
        """fresh tape: [(layers of slice k)] for k = 0..4"""
        out = []
        for k in range(5):
            ls = []
            for mod in getattr(self.lp.net, f"slice{k + 1}"):
                if isinstance(mod, torch.nn.MaxPool2d):
                    ls.append(_MaxPool())
                elif isinstance(mod, torch.nn.ReLU):
                    ls.append(_BNLReLU(None, {}, slope=0.0))
                else:
                    ls.append(_Conv(mod, {}, need_wgrad=False))
            out.append(_Seq(ls))
        return out

    def _scaled(self, x_nhwc):
        B, H, W, ld = x_nhwc.shape
        y = torch.empty((B, H, W, 32), device=x_nhwc.device, dtype=torch.float32)
        check(_lib.load().sgam_channel_affine_f32(_p(x_nhwc), ld, _p(y), 32, B * H * W, 3, self.shift, self.inv_scale, _stream()),
              "sgam_channel_affine_f32")
        return y

    def loss_and_grad(self, rec_nhwc, target_nhwc, grad_scale, values_only=False):
        """rec / target (B,H,W,>=3) NHWC (RGB first) -> (values per image [B] host floats, grad_scale * d sum_b val_b / d rec
        as (B,H,W,32) with the RGB channels filled; None with values_only: the forward of LPIPS.forward)"""
        lib = _lib.load()
        B, H, W, _ = rec_nhwc.shape
        t0, t1 = self._trunk(), self._trunk()
        f0, f1 = self._scaled(rec_nhwc), self._scaled(target_nhwc)
        vals = [0.0] * B
        dfeat = []
        for k in range(5):
            f0, f1 = t0[k].fwd(f0), t1[k].fwd(f1)
            Bk, Hk, Wk, Ck = f0.shape
            nblk = (Hk * Wk + 255) // 256
            part = torch.empty((Bk, nblk), device=f0.device, dtype=torch.float64)
            df = torch.empty_like(f0)
            check(lib.sgam_lpips_level_f32(_p(f0), _p(f1), _p(self.lp.lin_weight(k)), _p(part), _p(df), Bk, Hk * Wk, Ck, 1e-10,
                                           float(grad_scale), _stream()), "sgam_lpips_level_f32")
            ph = part.cpu().numpy().sum(axis=1) / (Hk * Wk)
            vals = [v + float(a) for v, a in zip(vals, ph)]
            dfeat.append(df)
        if values_only:
            return vals, None
        # backward through the trunk of the reconstruction: the five level gradients enter at their depths
        g = dfeat[4]
        for k in (4, 3, 2, 1, 0):
            g = t0[k].bwd(g)
            if k > 0:
                g = _axpby(g, dfeat[k - 1])
        dx = torch.empty((B, H, W, 32), device=g.device, dtype=torch.float32)
        check(lib.sgam_channel_affine_f32(_p(g), g.shape[3], _p(dx), 32, B * H * W, 3, self.zero, self.inv_scale, _stream()),
              "sgam_channel_affine_f32")
        return vals, dx


class _BNLReLU:
    """[BatchNorm2d in training mode ->] LeakyReLU(0.2) of the PatchGAN (discriminator/model.py:40-60)"""

    def __init__(self, bn, grads, slope=0.2, inference=False):            # slope 0: plain ReLU (the VGG16 trunk of LPIPS)
        self.bn, self.grads, self.slope = bn, grads, slope
        self.inference = inference                       # forward only (module.forward): an eval()-mode BatchNorm uses its running statistics

    def fwd(self, x):
        lib = _lib.load()
        B, H, W, C = x.shape
        rows = B * H * W
        self.x, self.mr = x, None
        y = torch.empty_like(x)
        if self.bn is not None:
            bn = self.bn
            if not bn.training and self.inference:
                # eval(): y = (x - running_mean) / sqrt(running_var + eps) * gamma + beta, nothing mutated (torch BatchNorm2d.forward)
                self.mr = torch.stack([bn.running_mean.float(), torch.rsqrt(bn.running_var.float() + bn.eps)], dim=1).contiguous()
                check(lib.sgam_bn_lrelu_fwd_f32(_p(x), _p(self.mr), _p(bn.weight.data), _p(bn.bias.data), _p(y), rows, C, self.slope, _stream()),
                      "sgam_bn_lrelu_fwd_f32")
                return y
            if not bn.training:
                # eval-mode BatchNorm normalises with the running statistics (and its backward differs): not built — the
                # reference trains the PatchGAN in train() mode (Lightning's fit); do not silently use batch statistics
                raise NotImplementedError("training through an eval()-mode BatchNorm2d: call discriminator.train() first")
            nb = lib.sgam_batchnorm_workspace_bytes(rows, C)
            ws = torch.empty((nb,), device=x.device, dtype=torch.uint8)
            self.mr = torch.empty((C, 2), device=x.device, dtype=torch.float32)
            track = bn.training and bn.track_running_stats
            check(lib.sgam_batchnorm_stats_f32(_p(x), _p(self.mr), _p(bn.running_mean) if track else None,
                                               _p(bn.running_var) if track else None, rows, C, bn.eps, bn.momentum, _p(ws), nb, _stream()),
                  "sgam_batchnorm_stats_f32")
            if track:
                bn.num_batches_tracked += 1
            check(lib.sgam_bn_lrelu_fwd_f32(_p(x), _p(self.mr), _p(bn.weight.data), _p(bn.bias.data), _p(y), rows, C, self.slope, _stream()),
                  "sgam_bn_lrelu_fwd_f32")
        else:
            check(lib.sgam_bn_lrelu_fwd_f32(_p(x), None, None, None, _p(y), rows, C, self.slope, _stream()), "sgam_bn_lrelu_fwd_f32")
        return y

    def bwd(self, dy, need_pgrad=True):
        lib = _lib.load()
        x = self.x
        B, H, W, C = x.shape
        rows = B * H * W
        nb = lib.sgam_batchnorm_workspace_bytes(rows, C)
        ws = torch.empty((nb,), device=x.device, dtype=torch.uint8)
        dx = torch.empty_like(x)
        if self.bn is None:
            check(lib.sgam_bn_lrelu_bwd_f32(_p(x), _p(dy), None, None, None, _p(dx), None, None, _p(dx), None, rows, C, self.slope, _p(ws),
                                            nb, _stream()), "sgam_bn_lrelu_bwd_f32")
            return dx
        bn = self.bn
        dg, db = torch.empty((C,), device=x.device), torch.empty((C,), device=x.device)
        gbuf, means = torch.empty_like(x), torch.empty((C, 2), device=x.device)
        check(lib.sgam_bn_lrelu_bwd_f32(_p(x), _p(dy), _p(self.mr), _p(bn.weight.data), _p(bn.bias.data), _p(dx), _p(dg), _p(db), _p(gbuf),
                                        _p(means), rows, C, self.slope, _p(ws), nb, _stream()), "sgam_bn_lrelu_bwd_f32")
        if need_pgrad:
            _accumulate(self.grads, bn.weight, dg)
            _accumulate(self.grads, bn.bias, db)
        return dx


def _accumulate(grads, p, g):
    grads[p] = g if p not in grads else _axpby(grads[p], g)


class _DiscTape:
    """one forward of NLayerDiscriminator.main on an NHWC batch (channels padded to 32) with everything the backward needs"""

    def __init__(self, disc, grads, inference=False):
        self.grads = grads
        self.layers = []
        mods = list(disc.main)
        i = 0
        while i < len(mods):
            conv = mods[i]
            bn = mods[i + 1] if i + 1 < len(mods) and isinstance(mods[i + 1], torch.nn.BatchNorm2d) else None
            has_act = any(isinstance(m, torch.nn.LeakyReLU) for m in mods[i + 1:i + 3])
            self.layers.append((_Conv(conv, {}), _BNLReLU(bn, grads, inference=inference) if has_act else None))
            i += 1 + (1 if bn is not None else 0) + (1 if has_act else 0)

    def fwd(self, x):
        for conv, act in self.layers:
            x = conv.fwd(x)
            if act is not None:
                x = act.fwd(x)
        return x                                              # logits (B,h,w,1)

    def bwd(self, dlogits, need_pgrad):
        """dlogits (B,h,w,32) (column 0 real) -> gradient w.r.t. the input (B,H,W,32); parameter gradients ACCUMULATE into grads"""
        dy = dlogits
        for conv, act in reversed(self.layers):
            if act is not None:
                dy = act.bwd(dy, need_pgrad)
            conv.need_wgrad = need_pgrad
            conv.grads.clear()
            dy = conv.bwd(_pad_channels(dy, conv.cout_k))
            for p, g in conv.grads.items():
                _accumulate(self.grads, p, g)
        return dy


class VQGANTrainer(AutoencoderTrainer):
    """`VQModel.training_step` (model.py:271-345) with `VQLPIPSWithDiscriminator` (vqperceptual.py:34-137) at
    perceptual_weight = 0: the autoencoder update with the adaptive-weighted generator term, then the discriminator update
    (hinge loss, its own Adam).  `loss_cfg` = a modules.losses.vqperceptual.VQLPIPSWithDiscriminator container."""

    def __init__(self, model, loss_cfg, phase=None, lr=None, process_group=None):
        super().__init__(model, phase=phase, lr=lr, codebook_weight=loss_cfg.codebook_weight, process_group=process_group)
        self.cfg, self.disc = loss_cfg, loss_cfg.discriminator
        self.lpips = _Lpips(loss_cfg.perceptual_loss) if loss_cfg.perceptual_weight > 0 else None
        # the discriminator is initialised by weights_init from each process's own RNG: without this every rank would train a
        # different PatchGAN on averaged gradients
        self.synced_tensors += self._broadcast_module(loss_cfg.discriminator)
        self.dgrads, self.dstate = {}, {}

    def _disc_factor(self):
        return self.cfg.disc_factor if self.global_step >= self.cfg.discriminator_iter_start else 0.0       # adopt_weight, :14-17

    def _hinge(self, logits, mode, gscale):
        lib = _lib.load()
        n = logits.numel()
        grad = torch.empty((n,), device=logits.device, dtype=torch.float32)
        part = torch.empty(((n + 255) // 256,), device=logits.device, dtype=torch.float64)
        check(lib.sgam_hinge_terms_f32(_p(logits), _p(grad), _p(part), n, mode, float(gscale), _stream()), "sgam_hinge_terms_f32")
        return _host_sum(part) / n, _pad_channels(grad.reshape(logits.shape), 32)

    def _norm(self, t):
        part = torch.empty(((t.numel() + 255) // 256,), device=t.device, dtype=torch.float64)
        check(_lib.load().sgam_sumsq_partial_f32(_p(t), _p(part), t.numel(), _stream()), "sgam_sumsq_partial_f32")
        return _host_sum(part) ** 0.5

    def step(self, x, x_dst, extrapolation_mask=None):
        cfg, lib = self.cfg, _lib.load()
        disc_factor = self._disc_factor()
        with _mfma_mode():
            # ---- optimizer_idx 0 (vqperceptual.py:77-110)
            fw = self._forward(x, x_dst, extrapolation_mask)
            rec, drec_nll = fw["rec"], fw["drec"]
            p_loss = 0.0
            if self.lpips is not None:
                # rec_loss = |x - xrec| + perceptual_weight * p_loss (p_loss (B,1,1,1) broadcast), nll = its mean (:79-89):
                # every image's LPIPS value enters the mean with weight 1 / B
                B = rec.shape[0]
                vals, dp = self.lpips.loss_and_grad(rec, ops.nchw_to_nhwc(x_dst), cfg.perceptual_weight / B)
                p_loss = sum(vals) / B
                fw["nll"] += cfg.perceptual_weight * p_loss
                drec_nll = _axpby(drec_nll, dp)
            tape_g = _DiscTape(self.disc, {})
            logits_fake = tape_g.fwd(_pad_channels(rec, 32))
            n_log = logits_fake.numel()
            mean_fake, dlog = self._hinge(logits_fake, 0, -1.0 / n_log)                # g_loss = -mean(logits_fake)
            g_loss = -mean_fake
            drec_g = tape_g.bwd(dlog, need_pgrad=False)
            # adaptive weight (:63-75): the two gradients w.r.t. the last layer's weight
            last = self.dec.layers[-1]
            keep = last.need_wgrad
            last.need_wgrad, norms = True, []
            for d in (drec_nll, drec_g):
                last.bwd(d, need_dx=False)
                norms.append(self._norm(self.grads[last.conv.weight]))
            last.need_wgrad = keep
            self.grads.clear()
            d_weight = min(max(norms[0] / (norms[1] + 1e-4), 0.0), 1e4) * cfg.discriminator_weight
            drec = _axpby(drec_nll, drec_g, 1.0, d_weight * disc_factor) if disc_factor != 0 else drec_nll
            self._backward(fw, drec)
            self.allreduce_grads()
            # ---- optimizer_idx 1 (:112-129): the discriminator sees the reconstruction of BEFORE the autoencoder update
            tape_r, tape_f = _DiscTape(self.disc, self.dgrads), _DiscTape(self.disc, self.dgrads)
            self.dgrads.clear()
            logits_real = tape_r.fwd(ops.nchw_to_nhwc(x_dst, c_pad=32))
            logits_fake2 = tape_f.fwd(_pad_channels(rec, 32))
            dm = 2 if getattr(cfg, "disc_loss_name", "hinge") == "vanilla" else 1        # softplus(-/+ l) instead of relu(1 -/+ l)
            lr_mean, dl_real = self._hinge(logits_real, -dm, 0.5 * disc_factor / logits_real.numel())
            lf_mean, dl_fake = self._hinge(logits_fake2, +dm, 0.5 * disc_factor / logits_fake2.numel())
            d_loss = disc_factor * 0.5 * (lr_mean + lf_mean)
            if disc_factor != 0:
                tape_r.bwd(dl_real, need_pgrad=True)
                tape_f.bwd(dl_fake, need_pgrad=True)
            self._allreduce(self.dgrads, list(self.disc.parameters()))          # the discriminator sits inside the DDP module too
        self.adam_step()                                        # opt_ae.step(); also advances global_step
        self._adam(list(self.disc.parameters()), self.dgrads, self.dstate)       # opt_disc.step()
        _invalidate_packs(self.disc)
        ae = fw["nll"] + d_weight * disc_factor * g_loss + self.codebook_weight * fw["qloss"]
        log = {"train/total_loss": ae, "train/quant_loss": fw["qloss"], "train/rec_loss": fw["nll"], "train/p_loss": p_loss, "train/d_weight": d_weight,
               "train/disc_factor": disc_factor, "train/g_loss": g_loss, "train/disc_loss": d_loss,
               "train/logits_real": self._mean_logit(logits_real), "train/logits_fake": self._mean_logit(logits_fake2)}
        return ae, log

    def _mean_logit(self, logits):
        return self._hinge(logits, 0, 0.0)[0]


def _invalidate_packs(model):
    """the packed / split / fragment-ordered copies of the weights are cached per (storage, version): an optimiser step
    through the raw pointer does not bump the version, so the cache keys are dropped explicitly (Conv2d._packed,
    AttnBlock._packed_qkv, VectorQuantizer2._codebook), and so are the captured graphs, which hold those copies"""
    for mod in model.modules():
        for key in ("_pack_key", "_qkv_key", "_cb_key"):
            if hasattr(mod, key):
                setattr(mod, key, None)
    if hasattr(model, "_graphs"):
        model._graphs.clear()


@contextlib.contextmanager
def _mfma_mode():
    old = ops.F32_MODE
    ops.set_f32_mode("mfma")
    try:
        yield
    finally:
        ops.set_f32_mode(old)
