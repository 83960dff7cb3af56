"""TEST INFRASTRUCTURE: the PyTorch-ROCm graph of a deepinv_amd.models.DRUNet (the same nn.Conv* modules evaluated by
ATen / MIOpen with autograd) - an independent GPU reference for the hand-written HIP forward / backward.  The product
has no such path (models/drunet.py raises for what its kernels do not cover)."""
import contextlib

import torch

from deepinv_amd.models.drunet import run_four_windows, run_replicate_padded


def forward_unet_torch(model, x0):
    """deepinv/models/drunet.py:200-210"""
    x1 = model.m_head(x0)
    x2 = model.m_down1(x1)
    x3 = model.m_down2(x2)
    x4 = model.m_down3(x3)
    x = model.m_body(x4)
    x = model.m_up3(x + x4)
    x = model.m_up2(x + x3)
    x = model.m_up1(x + x2)
    return model.m_tail(x + x1)


def torch_forward(model, x, sigma):
    """DRUNet.forward (drunet.py:212-263) through the PyTorch graph: noise map, padding rules, U-Net"""
    run = lambda inp: forward_unet_torch(model, inp)
    xin = torch.cat((x, model._noise_map(x, sigma)), 1)
    if all(s % 8 == 0 and s > 31 for s in xin.shape[2:]):
        return run(xin)
    if model.training or any(xin.size(2 + i) < 64 for i in range(model.dim)):
        return run_replicate_padded(run, xin, multiple=16)
    return run_four_windows(run, xin, field=64)


@contextlib.contextmanager
def torch_backend(model):
    """inside the block `model(x, sigma)` evaluates the PyTorch graph (e.g. a denoiser buried in an unfolded network)"""
    model.forward = lambda x, sigma: torch_forward(model, x, sigma)
    try:
        yield model
    finally:
        del model.forward
