"""ORACLE (test infrastructure only) — functional torch-CPU fp32 restatement of the
conditional VQGAN forward of the reference, driven by a plain ``state_dict`` with the
reference's key names.  It is the parity checker for the HIP path and is never imported
by the product package.

Reference:
  VQModel.encode/decode/forward  sgam/generative_sensing_module/model.py:106-167
  Encoder / Decoder              .../modules/diffusionmodules/model.py:342-433 / 437-539
  ResnetBlock / AttnBlock        .../modules/diffusionmodules/model.py:78-137 / 140-192
  Up/Downsample, Normalize       .../modules/diffusionmodules/model.py:34-75
  VectorQuantizer2.forward       .../modules/vqvae/quantize.py:275-319
  get_multiple_codewords         .../modules/vqvae/quantize.py:344-381
"""
import torch
import torch.nn.functional as F


def _swish(x):
    return x * torch.sigmoid(x)


def _norm(sd, p, x):
    return F.group_norm(x, 32, sd[p + ".weight"], sd[p + ".bias"], eps=1e-6)


def _conv(sd, p, x, stride=1, padding=0):
    return F.conv2d(x, sd[p + ".weight"], sd[p + ".bias"], stride=stride, padding=padding)


def resnet_block(sd, p, x):
    """model.py:117-137 with temb=None, dropout p=0."""
    h = _conv(sd, p + ".conv1", _swish(_norm(sd, p + ".norm1", x)), padding=1)
    h = _conv(sd, p + ".conv2", _swish(_norm(sd, p + ".norm2", h)), padding=1)
    if (p + ".nin_shortcut.weight") in sd:
        x = _conv(sd, p + ".nin_shortcut", x)
    return x + h


def attn_block(sd, p, x):
    """model.py:168-192: single-head spatial self-attention, softmax over keys."""
    h = _norm(sd, p + ".norm", x)
    q, k, v = _conv(sd, p + ".q", h), _conv(sd, p + ".k", h), _conv(sd, p + ".v", h)
    b, c, hh, ww = q.shape
    q = q.reshape(b, c, hh * ww).permute(0, 2, 1)
    k = k.reshape(b, c, hh * ww)
    w_ = torch.bmm(q, k) * (int(c) ** (-0.5))
    w_ = F.softmax(w_, dim=2)
    v = v.reshape(b, c, hh * ww)
    h = torch.bmm(v, w_.permute(0, 2, 1)).reshape(b, c, hh, ww)
    return x + _conv(sd, p + ".proj_out", h)


def _levels(dd):
    return len(dd["ch_mult"]), dd["num_res_blocks"]


def _has(sd, key):
    return key in sd


def encoder(sd, dd, x, p="encoder"):
    """model.py:405-433.  Attention placement is whatever the state_dict holds
    (it derives from ddconfig.resolution, not from the input size)."""
    nlev, nres = _levels(dd)
    h = _conv(sd, p + ".conv_in", x, padding=1)
    for lv in range(nlev):
        for ib in range(nres):
            h = resnet_block(sd, f"{p}.down.{lv}.block.{ib}", h)
            if _has(sd, f"{p}.down.{lv}.attn.{ib}.norm.weight"):
                h = attn_block(sd, f"{p}.down.{lv}.attn.{ib}", h)
        if lv != nlev - 1:
            h = F.pad(h, (0, 1, 0, 1), mode="constant", value=0)
            h = _conv(sd, f"{p}.down.{lv}.downsample.conv", h, stride=2)
    h = resnet_block(sd, p + ".mid.block_1", h)
    h = attn_block(sd, p + ".mid.attn_1", h)
    h = resnet_block(sd, p + ".mid.block_2", h)
    h = _swish(_norm(sd, p + ".norm_out", h))
    return _conv(sd, p + ".conv_out", h, padding=1)


def decoder(sd, dd, z, p="decoder"):
    """model.py:508-539."""
    nlev, nres = _levels(dd)
    h = _conv(sd, p + ".conv_in", z, padding=1)
    h = resnet_block(sd, p + ".mid.block_1", h)
    h = attn_block(sd, p + ".mid.attn_1", h)
    h = resnet_block(sd, p + ".mid.block_2", h)
    for lv in reversed(range(nlev)):
        for ib in range(nres + 1):
            h = resnet_block(sd, f"{p}.up.{lv}.block.{ib}", h)
            if _has(sd, f"{p}.up.{lv}.attn.{ib}.norm.weight"):
                h = attn_block(sd, f"{p}.up.{lv}.attn.{ib}", h)
        if lv != 0:
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")
            h = _conv(sd, f"{p}.up.{lv}.upsample.conv", h, padding=1)
    h = _swish(_norm(sd, p + ".norm_out", h))
    return _conv(sd, p + ".conv_out", h, padding=1)


def distances(sd, z_flat):
    """quantize.py:285-287, exact expression order."""
    e = sd["quantize.embedding.weight"]
    return torch.sum(z_flat ** 2, dim=1, keepdim=True) + torch.sum(e ** 2, dim=1) - 2 * \
        torch.einsum("bd,dn->bn", z_flat, e.permute(1, 0))


def quantize(sd, z):
    """VectorQuantizer2.forward at inference (quantize.py:275-319): returns z_q (NCHW, incl.
    the straight-through expression z + (z_q - z)), indices (B,h,w) int64, distance matrix, commitment loss."""
    e = sd["quantize.embedding.weight"]
    zp = z.permute(0, 2, 3, 1).contiguous()
    zf = zp.view(-1, e.shape[1])
    d = distances(sd, zf)
    idx = torch.argmin(d, dim=1)
    z_q = F.embedding(idx, e).view(zp.shape)
    # :296-304, legacy form, beta = 0.25 (the detach()es only matter to autograd: tests/test_gpu_training.py differentiates
    # this restatement to check the HIP backward pass)
    loss = torch.mean((z_q.detach() - zp) ** 2) + 0.25 * torch.mean((z_q - zp.detach()) ** 2)
    z_q = zp + (z_q - zp).detach()
    return z_q.permute(0, 3, 1, 2).contiguous(), idx.view(zp.shape[:-1]), d, loss


def get_multiple_codewords(sd, z, topk, sample_number, extrapolation_mask):
    """quantize.py:344-381 — including its quirks: batch 1 / 16x16 only, row 0's
    distribution used for every token (:358), CPU RNG stream, pure gather (no
    straight-through add)."""
    e = sd["quantize.embedding.weight"]
    em = F.interpolate(extrapolation_mask.float(), size=(16, 16))
    zp = z.permute(0, 2, 3, 1).contiguous()
    zf = zp.view(-1, e.shape[1])
    d = distances(sd, zf)
    tk = torch.topk(d, topk, dim=1, largest=False)
    dist = F.softmax(-tk.values / 1, dim=-1)
    rows = []
    for i in range(dist.shape[0]):
        rows.append(tk.indices[i][torch.multinomial(dist[0], sample_number, replacement=True)])
    sampled = torch.stack(rows)
    keep = (1 - em.view(-1, 1))[:, 0] != 0
    sampled[keep] = tk.indices[keep, :1].expand(-1, sample_number)
    z_qs = F.embedding(sampled.flatten(), e).view(1, *zp.shape[1:3], sample_number, zp.shape[-1])
    z_qs = z_qs.permute(0, 3, 4, 1, 2).contiguous()
    idx = sampled.flatten().view(1, *zp.shape[1:3], sample_number).permute(0, 3, 1, 2)
    return z_qs, idx


def encode_features(sd, dd, x, extrapolation_mask=None):
    """VQModel.encode up to quant_conv (model.py:106-116)."""
    if extrapolation_mask is None:
        extrapolation_mask = torch.zeros([x.shape[0], 1, *x.shape[2:]])
    x = torch.cat([x, extrapolation_mask], 1)
    x = _conv(sd, "conv_in", x)
    return _conv(sd, "quant_conv", encoder(sd, dd, x))


def decode(sd, dd, quant):
    """VQModel.decode (model.py:131-134)."""
    return decoder(sd, dd, _conv(sd, "post_quant_conv", quant))


@torch.no_grad()
def forward(sd, dd, x, extrapolation_mask=None, topk=None, sample_number=1):
    """VQModel.forward (model.py:141-167) -> dict(dec, indices, pre_quant, quant)."""
    pre = encode_features(sd, dd, x, extrapolation_mask)
    if topk is None:
        quant, idx, _, loss = quantize(sd, pre)
        return {"dec": decode(sd, dd, quant), "indices": idx, "pre_quant": pre, "quant": quant, "emb_loss": loss}
    quants, idx = get_multiple_codewords(sd, pre, topk, sample_number, extrapolation_mask)
    decs = [decode(sd, dd, quants[:, i])[None] for i in range(sample_number)]
    return {"dec": decs, "indices": idx, "pre_quant": pre, "quant": quants}
