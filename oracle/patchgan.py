"""ORACLE (test infrastructure only) — functional torch-CPU restatement of the reference's PatchGAN discriminator and of
the GAN terms of its loss, driven by a plain state_dict with the reference's key names; never imported by the product.

Reference:
  NLayerDiscriminator.forward     sgam/generative_sensing_module/modules/discriminator/model.py:17-67
  hinge_d_loss, adaptive weight,
  VQLPIPSWithDiscriminator.forward sgam/generative_sensing_module/modules/losses/vqperceptual.py:17-21, 63-129
Pinned by tests/test_oracle_golden.py against tests/golden/train_step_gan_small.npz (the reference's own numbers)."""
import torch
import torch.nn.functional as F


def discriminator(sd, x, n_layers=3, training=True, momentum=0.1, prefix="main."):
    """logits of NLayerDiscriminator(n_layers, kernel 4); BatchNorm in training mode uses (and updates, in place in `sd`) batch
    statistics exactly like nn.BatchNorm2d"""
    i = 0
    h = F.leaky_relu(F.conv2d(x, sd[f"{prefix}{i}.weight"], sd[f"{prefix}{i}.bias"], stride=2, padding=1), 0.2)
    i += 2
    for n in range(1, n_layers + 1):
        stride = 2 if n < n_layers else 1
        h = F.conv2d(h, sd[f"{prefix}{i}.weight"], None, stride=stride, padding=1)
        bn = f"{prefix}{i + 1}."
        h = F.batch_norm(h, sd[bn + "running_mean"], sd[bn + "running_var"], sd[bn + "weight"], sd[bn + "bias"], training, momentum, 1e-5)
        if training and (bn + "num_batches_tracked") in sd:
            sd[bn + "num_batches_tracked"] = sd[bn + "num_batches_tracked"] + 1          # nn.BatchNorm2d's counter
        h = F.leaky_relu(h, 0.2)
        i += 3
    return F.conv2d(h, sd[f"{prefix}{i}.weight"], sd[f"{prefix}{i}.bias"], stride=1, padding=1)


def hinge_d_loss(logits_real, logits_fake):
    return 0.5 * (torch.mean(F.relu(1.0 - logits_real)) + torch.mean(F.relu(1.0 + logits_fake)))


def adaptive_weight(nll_loss, g_loss, last_layer, disc_weight):
    nll_grads = torch.autograd.grad(nll_loss, last_layer, retain_graph=True)[0]
    g_grads = torch.autograd.grad(g_loss, last_layer, retain_graph=True)[0]
    return (torch.clamp(torch.norm(nll_grads) / (torch.norm(g_grads) + 1e-4), 0.0, 1e4).detach() * disc_weight)
