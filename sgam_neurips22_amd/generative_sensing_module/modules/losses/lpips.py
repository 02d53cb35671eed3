"""Parameter container of the reference's LPIPS (sgam/generative_sensing_module/modules/losses/lpips.py:10-123): the VGG16
trunk sliced at relu1_2 / relu2_2 / relu3_3 / relu4_3 / relu5_3, the input ScalingLayer and the five 1x1 `lin` layers, with the
reference's state_dict keys (`net.slice<k>.<i>.weight`, `lin<k>.model.1.weight`, `scaling_layer.shift / scale`).  The
arithmetic (forward of both images, backward w.r.t. the reconstruction) runs in sgam_neurips22_amd/training.py on the HIP
kernels.  Weights: the `lin` layers ship with the reference (modules/autoencoder/lpips/vgg.pth); the VGG16 trunk is
torchvision's ImageNet checkpoint, which this repository cannot fetch — `load_state_dict` whatever checkpoint you have."""
import torch
import torch.nn as nn

from ..diffusionmodules.model import Conv2d

VGG16_CFG = [64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512, "M"]   # torchvision cfg "D"
SLICES = [(0, 4), (4, 9), (9, 16), (16, 23), (23, 30)]                                                # lpips.py:86-95


class _Holder(nn.Module):
    pass


class LPIPS(nn.Module):
    def __init__(self, use_dropout=True):
        super().__init__()
        self.chns = [64, 128, 256, 512, 512]
        self.scaling_layer = _Holder()
        self.scaling_layer.register_buffer("shift", torch.tensor([-.030, -.088, -.188])[None, :, None, None])
        self.scaling_layer.register_buffer("scale", torch.tensor([.458, .448, .450])[None, :, None, None])
        feats, cin = [], 3
        for v in VGG16_CFG:
            if v == "M":
                feats.append(nn.MaxPool2d(kernel_size=2, stride=2))
            else:
                feats += [Conv2d(cin, v, kernel_size=3, padding=1), nn.ReLU(inplace=True)]
                cin = v
        self.net = _Holder()
        for k, (a, b) in enumerate(SLICES):
            sl = nn.Sequential()
            for i in range(a, b):
                sl.add_module(str(i), feats[i])
            setattr(self.net, f"slice{k + 1}", sl)
        for k, c in enumerate(self.chns):
            lin = _Holder()
            layers = ([nn.Dropout()] if use_dropout else []) + [nn.Conv2d(c, 1, 1, stride=1, padding=0, bias=False)]
            lin.model = nn.Sequential(*layers)
            setattr(self, f"lin{k}", lin)
        for p in self.parameters():
            p.requires_grad = False

    def lin_weight(self, k):
        return getattr(self, f"lin{k}").model[-1].weight.detach().reshape(-1).float().contiguous()

    def forward(self, input, target):
        """reference lpips.py:41-55: (B,3,H,W) x 2 in [-1, 1] -> (B,1,1,1) perceptual distances — the forward half of the tape
        training.py differentiates (same kernels, strict-fp32 MFMA mode), for callers of `loss.perceptual_loss(a, b)`.  The
        result is DETACHED (no autograd graph; the gradient w.r.t. the reconstruction comes from training._Lpips.loss_and_grad),
        so inputs that require grad are refused rather than silently cut off."""
        from .... import ops, training
        if torch.is_grad_enabled() and (input.requires_grad or target.requires_grad):
            raise RuntimeError("LPIPS.forward returns a detached tensor: use training._Lpips(...).loss_and_grad for the gradient "
                               "(or call under torch.no_grad() / on detached inputs)")
        with training._mfma_mode():
            vals, _ = training._Lpips(self).loss_and_grad(ops.nchw_to_nhwc(input.float().contiguous(), c_pad=32),
                                                         ops.nchw_to_nhwc(target.float().contiguous(), c_pad=32), 0.0, values_only=True)
        return torch.tensor(vals, device=input.device, dtype=torch.float32).view(-1, 1, 1, 1)
