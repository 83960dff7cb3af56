"""DRUNet forward as plain functional torch-CPU code driven by a reference ``state_dict``.

TEST INFRASTRUCTURE.  Follows deepinv/models/drunet.py:200-263 (forward / forward_unet),
:403-434 (ResBlock 'CRC'), :493-552 (convtranspose up / strideconv down), utils.py:49-98
(test_pad / test_onesplit are only needed for shapes that are not multiples of 8).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def init_state_dict(in_channels=2, out_channels=2, nc=(64, 128, 256, 512), nb=4, seed=0, res_gain=None):
    """Random DRUNet weights exactly like the reference's ``weights_init_drunet`` (orthogonal, gain 0.2;
    drunet.py:689-692) in the module order of DRUNet.__init__ (drunet.py:56-153).

    ``res_gain`` (e.g. 1.0) replaces the gain of the 56 ResBlock convolutions only.  With the reference's 0.2 every
    ResBlock branch is ~0.04x its input, which hides the ResBlock kernels' rounding behind the identity path; an
    O(1) gain makes the end-to-end error a statement about those kernels."""
    g = torch.Generator().manual_seed(seed)
    sd = {}

    def conv(name, co, ci, k):
        w = torch.empty(co, ci, k, k)
        gain = res_gain if (res_gain is not None and ".res." in name) else 0.2
        torch.nn.init.orthogonal_(w, gain=gain, generator=g)
        sd[name] = w

    conv("m_head.weight", nc[0], in_channels + 1, 3)
    for lvl, name in enumerate(("m_down1", "m_down2", "m_down3")):
        for b in range(nb):
            conv(f"{name}.{b}.res.0.weight", nc[lvl], nc[lvl], 3)
            conv(f"{name}.{b}.res.2.weight", nc[lvl], nc[lvl], 3)
        conv(f"{name}.{nb}.weight", nc[lvl + 1], nc[lvl], 2)
    for b in range(nb):
        conv(f"m_body.{b}.res.0.weight", nc[3], nc[3], 3)
        conv(f"m_body.{b}.res.2.weight", nc[3], nc[3], 3)
    for lvl, name in zip((3, 2, 1), ("m_up3", "m_up2", "m_up1")):
        conv(f"{name}.0.weight", nc[lvl], nc[lvl - 1], 2)  # ConvTranspose2d weight is [Cin, Cout, 2, 2]
        for b in range(1, nb + 1):
            conv(f"{name}.{b}.res.0.weight", nc[lvl - 1], nc[lvl - 1], 3)
            conv(f"{name}.{b}.res.2.weight", nc[lvl - 1], nc[lvl - 1], 3)
    conv("m_tail.weight", out_channels, nc[0], 3)
    return sd


def _res(sd, prefix, x):
    r = F.conv2d(x, sd[prefix + ".res.0.weight"], padding=1)
    r = F.relu(r)
    r = F.conv2d(r, sd[prefix + ".res.2.weight"], padding=1)
    return x + r


def forward_unet(sd, x0, nb=4):
    """drunet.py:200-210"""
    x1 = F.conv2d(x0, sd["m_head.weight"], padding=1)
    skips = [x1]
    x = x1
    for name in ("m_down1", "m_down2", "m_down3"):
        for b in range(nb):
            x = _res(sd, f"{name}.{b}", x)
        x = F.conv2d(x, sd[f"{name}.{nb}.weight"], stride=2)
        skips.append(x)
    for b in range(nb):
        x = _res(sd, f"m_body.{b}", x)
    for lvl, name in zip((3, 2, 1), ("m_up3", "m_up2", "m_up1")):
        x = F.conv_transpose2d(x + skips[lvl], sd[f"{name}.0.weight"], stride=2)
        for b in range(1, nb + 1):
            x = _res(sd, f"{name}.{b}", x)
    return F.conv2d(x + skips[0], sd["m_tail.weight"], padding=1)


def drunet(sd, x, sigma, nb=4):
    """DRUNet.forward for spatial sizes that are multiples of 8 and > 31 (drunet.py:212-263)."""
    if isinstance(sigma, torch.Tensor) and sigma.ndim > 0 and sigma.numel() > 1:
        m = sigma.reshape(-1, 1, 1, 1).expand(-1, 1, *x.shape[2:]) if sigma.numel() == x.shape[0] else sigma
    else:
        m = torch.full((x.shape[0], 1, *x.shape[2:]), float(sigma), dtype=x.dtype)
    assert all(s % 8 == 0 and s > 31 for s in x.shape[2:]), "oracle restates the shape-safe branch only"
    return forward_unet(sd, torch.cat((x, m), 1), nb)


def forward_unet_nd(sd, x0, nb, dim):
    """forward_unet (drunet.py:200-210) for dim = 2 or 3 (Conv{dim}d / ConvTranspose{dim}d) with the reference's
    state_dict naming, including its `sequential` rule (drunet.py:279-297): a stage made of ONE module is that module
    itself, so with nb = 1 the body's keys are m_body.res.{0,2}.weight"""
    conv = {2: F.conv2d, 3: F.conv3d}[dim]
    convt = {2: F.conv_transpose2d, 3: F.conv_transpose3d}[dim]

    def res(prefix, x):
        r = F.relu(conv(x, sd[prefix + ".res.0.weight"], padding=1))
        return x + conv(r, sd[prefix + ".res.2.weight"], padding=1)

    x1 = conv(x0, sd["m_head.weight"], padding=1)
    skips = [x1]
    x = x1
    for name in ("m_down1", "m_down2", "m_down3"):
        for b in range(nb):
            x = res(f"{name}.{b}", x)
        x = conv(x, sd[f"{name}.{nb}.weight"], stride=2)
        skips.append(x)
    for b in range(nb):
        x = res("m_body" if nb == 1 else f"m_body.{b}", x)
    for lvl, name in zip((3, 2, 1), ("m_up3", "m_up2", "m_up1")):
        x = convt(x + skips[lvl], sd[f"{name}.0.weight"], stride=2)
        for b in range(1, nb + 1):
            x = res(f"{name}.{b}", x)
    return conv(x + skips[0], sd["m_tail.weight"], padding=1)
