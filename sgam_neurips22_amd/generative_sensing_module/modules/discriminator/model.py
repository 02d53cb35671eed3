"""PatchGAN discriminator of the reference's loss (sgam/generative_sensing_module/modules/discriminator/model.py:17-67, the
pix2pix NLayerDiscriminator): the parameter container with the reference's state_dict keys (`main.<i>.weight` ...); the
arithmetic runs in sgam_neurips22_amd/training.py (forward with tape, backward) on the HIP kernels — SURVEY §8 f4."""
import torch
import torch.nn as nn

from ..diffusionmodules.model import Conv2d


def weights_init(m):
    """reference :8-14: Conv weights ~ N(0, 0.02), BatchNorm weight ~ N(1, 0.02), bias 0"""
    name = m.__class__.__name__
    if name.find("Conv") != -1:
        nn.init.normal_(m.weight.data, 0.0, 0.02)
    elif name.find("BatchNorm") != -1:
        nn.init.normal_(m.weight.data, 1.0, 0.02)
        nn.init.constant_(m.bias.data, 0)


class NLayerDiscriminator(nn.Module):
    def __init__(self, input_nc=3, ndf=64, n_layers=3, kernel_width=4, use_actnorm=False):
        super().__init__()
        if use_actnorm:
            raise NotImplementedError("ActNorm discriminator (use_actnorm=True) is not built; the shipped configs use BatchNorm2d")
        kw, padw = kernel_width, 1
        seq = [Conv2d(input_nc, ndf, kernel_size=kw, stride=2, padding=padw), nn.LeakyReLU(0.2, True)]
        nf_mult = 1
        for n in range(1, n_layers):
            nf_prev, nf_mult = nf_mult, min(2 ** n, 8)
            seq += [Conv2d(ndf * nf_prev, ndf * nf_mult, kernel_size=kw, stride=2, padding=padw, bias=False),
                    nn.BatchNorm2d(ndf * nf_mult), nn.LeakyReLU(0.2, True)]
        nf_prev, nf_mult = nf_mult, min(2 ** n_layers, 8)
        seq += [Conv2d(ndf * nf_prev, ndf * nf_mult, kernel_size=kw, stride=1, padding=padw, bias=False),
                nn.BatchNorm2d(ndf * nf_mult), nn.LeakyReLU(0.2, True)]
        seq += [Conv2d(ndf * nf_mult, 1, kernel_size=kw, stride=1, padding=padw)]
        self.main = nn.Sequential(*seq)

    def forward(self, input):
        """reference :65-67: (B,C,H,W) -> (B,1,h,w) patch logits on the HIP kernels, with torch's BatchNorm2d semantics per mode:
        train() normalises with batch statistics and updates the running statistics; eval() normalises with the running statistics
        and mutates nothing.  The result is DETACHED — no autograd graph is recorded here (gradients of this network are the
        business of training.VQGANTrainer's tape) — so an input that requires grad is refused rather than silently cut off."""
        from .... import ops, training
        if torch.is_grad_enabled() and input.requires_grad:
            raise RuntimeError("NLayerDiscriminator.forward returns a detached tensor: differentiate through training.VQGANTrainer "
                               "(or call under torch.no_grad() / on a detached input)")
        with training._mfma_mode():
            logits = training._DiscTape(self, {}, inference=True).fwd(ops.nchw_to_nhwc(input.detach().float().contiguous(), c_pad=32))
        return ops.nhwc_to_nchw(logits[..., :1].contiguous())
