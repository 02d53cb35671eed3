"""Hyper-parameter / parameter container of the reference's VQLPIPSWithDiscriminator
(sgam/generative_sensing_module/modules/losses/vqperceptual.py:34-137), consumed by sgam_neurips22_amd.training.VQGANTrainer.
`perceptual_loss` is the LPIPS container (modules/losses/lpips.py): its VGG16 trunk needs torchvision's ImageNet checkpoint, which
cannot be fetched here — load one before training with perceptual_weight > 0."""
import torch.nn as nn

from ..discriminator.model import NLayerDiscriminator, weights_init
from .lpips import LPIPS


class VQLPIPSWithDiscriminator(nn.Module):
    def __init__(self, disc_start, codebook_weight=1.0, pixelloss_weight=1.0, disc_num_layers=3, disc_in_channels=3, disc_factor=1.0,
                 disc_weight=1.0, perceptual_weight=1.0, use_actnorm=False, disc_conditional=False, disc_ndf=64, disc_loss="hinge",
                 use_discriminative_loss=False, disp_loss_weight=None, disc_update_every_n_step=None, kernel_width=4):
        super().__init__()
        if disc_loss not in ("hinge", "vanilla"):
            raise ValueError(f"Unknown GAN loss '{disc_loss}'.")
        if disc_conditional:
            # the reference's forward asserts `cond is not None` in this mode and its training_step never passes one (model.py:324):
            # a conditional discriminator cannot be trained through the reference's own step either
            raise NotImplementedError("disc_conditional: VQModel.training_step passes no `cond` (model.py:324-336)")
        self.disc_loss_name = disc_loss
        self.codebook_weight, self.pixel_weight, self.perceptual_weight = codebook_weight, pixelloss_weight, perceptual_weight
        self.use_discriminative_loss = use_discriminative_loss
        self.perceptual_loss = LPIPS().eval()
        self.discriminator = NLayerDiscriminator(input_nc=disc_in_channels, n_layers=disc_num_layers, use_actnorm=use_actnorm,
                                                 ndf=disc_ndf, kernel_width=kernel_width).apply(weights_init)
        self.discriminator_iter_start = disc_start
        self.disc_factor, self.discriminator_weight, self.disc_conditional = disc_factor, disc_weight, disc_conditional
