"""Blur / BlurFFT / Downsampling kernels vs the CPU oracle (fp32, 1e-4 relative)."""
import pytest
import torch

from conftest import dot_test, rel_err
from oracle import physics_cpu as O

pytestmark = pytest.mark.gpu
TOL = 1e-4
PADS = ["valid", "circular", "reflect", "replicate", "constant"]


def _g(s=0):
    return torch.Generator().manual_seed(s)


@pytest.mark.parametrize("padding", PADS)
@pytest.mark.parametrize("img,filt", [((2, 3, 17, 19), (1, 1, 5, 5)), ((1, 3, 9, 8), (1, 3, 5, 4)),
                                      ((2, 2, 16, 16), (2, 1, 6, 6)), ((1, 1, 32, 40), (1, 1, 9, 9))])
def test_conv2d_and_transpose(dev, padding, img, filt):
    import deepinv_amd as dinv

    g = _g()
    x = torch.randn(*img, generator=g)
    k = torch.rand(*filt, generator=g)
    k = k / k.sum(dim=(-2, -1), keepdim=True)
    phys = dinv.physics.Blur(filter=k, padding=padding, device=dev)
    y = phys.A(x.to(dev))
    y_ref = O.conv2d(x, k, padding)
    assert y.shape == y_ref.shape
    assert rel_err(y, y_ref) < TOL
    v = torch.randn(y_ref.shape, generator=g)
    xa = phys.A_adjoint(v.to(dev))
    assert rel_err(xa, O.conv_transpose2d(v, k, padding, img[2], img[3])) < TOL
    assert dot_test(phys, x.to(dev), y) < 1e-5


@pytest.mark.parametrize("padding", PADS)
@pytest.mark.parametrize("img,filt", [((2, 3, 17, 19), (1, 1, 5, 5)), ((2, 3, 12, 16), (1, 3, 5, 4)), ((2, 2, 16, 16), (2, 1, 6, 6)),
                                      ((3, 2, 20, 11), (3, 2, 3, 3))])
def test_filter_gradients_match_autograd_oracle(dev, padding, img, filt):
    """d/d(filter) of conv2d and conv_transpose2d (blind / learned kernels; the reference gets it from autograd through F.conv2d,
    convolution.py:42-164) against autograd through the CPU oracle: shared and per-sample / per-channel filters, every padding;
    and the image gradients through the same graph (a filter that requires grad no longer switches anything off)"""
    import deepinv_amd.physics.functional as dF

    g = _g(5)
    x = torch.randn(*img, generator=g)
    k = torch.rand(*filt, generator=g)
    y_shape = O.conv2d(x, k, padding).shape
    wy, wx = torch.randn(*y_shape, generator=g), torch.randn(*img, generator=g)
    yin = torch.randn(*y_shape, generator=g)

    def grads(conv, convT, to):
        xs, ks, ys = (t.clone().to(to).requires_grad_() for t in (x, k, yin))
        g1 = torch.autograd.grad((conv(xs, ks) * wy.to(to)).sum(), (xs, ks))
        g2 = torch.autograd.grad((convT(ys, ks) * wx.to(to)).sum(), (ys, ks))
        return [t.cpu() for t in (*g1, *g2)]

    ref = grads(lambda a, b: O.conv2d(a, b, padding), lambda a, b: O.conv_transpose2d(a, b, padding, img[2], img[3]), "cpu")
    got = grads(lambda a, b: dF.conv2d(a, b, padding=padding), lambda a, b: dF.conv_transpose2d(a, b, padding=padding), dev)
    for name, r, o in zip(("conv/x", "conv/filter", "convT/y", "convT/filter"), ref, got):
        assert o.shape == r.shape, name
        assert rel_err(o, r) < TOL, name


@pytest.mark.parametrize("padding", PADS)
@pytest.mark.parametrize("vol,filt", [((2, 2, 9, 17, 19), (1, 1, 3, 5, 5)), ((1, 3, 8, 12, 10), (1, 3, 4, 3, 2)), ((2, 1, 16, 16, 16), (2, 1, 3, 3, 3))])
def test_volume_blur(dev, padding, vol, filt):
    """Blur on 5-D tensors (blur.py:535-561 -> conv3d / conv_transpose3d and their FFT forms, convolution.py:333-640) against the
    CPU oracle, the dot test, spatial == FFT, and the filter gradient against autograd through the oracle"""
    import deepinv_amd as dinv
    import deepinv_amd.physics.functional as dF

    g = _g(9)
    x = torch.randn(*vol, generator=g)
    k = torch.rand(*filt, generator=g)
    phys = dinv.physics.Blur(filter=k, padding=padding, device=dev)
    y = phys.A(x.to(dev))
    y_ref = O.conv3d(x, k, padding)
    assert y.shape == y_ref.shape and rel_err(y, y_ref) < TOL
    v = torch.randn(y_ref.shape, generator=g)
    assert rel_err(phys.A_adjoint(v.to(dev)), O.conv_transpose3d(v, k, padding, vol[2:])) < TOL
    assert dot_test(phys, x.to(dev), y) < 1e-5
    pf = dinv.physics.Blur(filter=k, padding=padding, use_fft=True, device=dev)
    assert rel_err(pf.A(x.to(dev)), y_ref) < TOL and rel_err(pf.A_adjoint(v.to(dev)), phys.A_adjoint(v.to(dev))) < TOL
    ks, kd = k.clone().requires_grad_(), k.clone().to(dev).requires_grad_()
    (O.conv3d(x, ks, padding) * v).sum().backward()
    (dF.conv3d(x.to(dev), kd, padding=padding) * v.to(dev)).sum().backward()
    assert rel_err(kd.grad, ks.grad) < TOL


@pytest.mark.parametrize("padding", PADS)
def test_blur_fft_path_matches_spatial(dev, padding):
    """reference test_physics_functional.py:158-246 (spatial == FFT implementation)"""
    import deepinv_amd.physics.functional as dF

    g = _g(1)
    x = torch.randn(2, 3, 17, 12, generator=g).to(dev)
    k = torch.rand(1, 1, 5, 4, generator=g).to(dev)
    assert rel_err(dF.conv2d_fft(x, k, padding=padding), dF.conv2d(x, k, padding=padding)) < TOL
    y = dF.conv2d(x, k, padding=padding)
    assert rel_err(dF.conv_transpose2d_fft(y, k, padding=padding), dF.conv_transpose2d(y, k, padding=padding)) < TOL


@pytest.mark.parametrize("img_size,fsize", [((3, 17, 19), (7, 7)), ((1, 16, 16), (4, 4)), ((3, 256, 256), (9, 9)),
                                            ((2, 15, 20), (5, 6))])
def test_blurfft(dev, img_size, fsize):
    import deepinv_amd as dinv

    g = _g(2)
    x = torch.rand(2, *img_size, generator=g)
    k = dinv.physics.functional.gaussian_blur(psf_size=fsize, sigma=(2.0, 2.0)) if fsize[0] == fsize[1] else \
        torch.rand(1, 1, *fsize, generator=g)
    k = k / k.sum()
    phys = dinv.physics.BlurFFT(img_size=img_size, filter=k, device=dev)
    mask, angle = O.blurfft_params(img_size, k)
    assert rel_err(phys.mask, mask) < TOL
    y = phys.A(x.to(dev))
    y_ref = O.blurfft_A(x, mask, angle, img_size)
    assert rel_err(y, y_ref) < TOL
    # BlurFFT == circular Blur (reference test_physics.py:1338-1381, atol 1e-5)
    yb = dinv.physics.Blur(filter=k, padding="circular", device=dev).A(x.to(dev))
    assert torch.allclose(y, yb, atol=1e-5)
    assert rel_err(phys.A_adjoint(y), O.blurfft_AT(y_ref, mask, angle, img_size)) < TOL
    assert dot_test(phys, x.to(dev), y) < 1e-5
    z = torch.rand(2, *img_size, generator=g)
    assert rel_err(phys.prox_l2(z.to(dev), y, 1.3), O.blurfft_prox_l2(z, y_ref, 1.3, mask, angle, img_size)) < TOL


@pytest.mark.parametrize("filt", ["bicubic", "bilinear", "gaussian", None])
@pytest.mark.parametrize("factor,img_size,padding", [(2, (3, 16, 24), "circular"), (4, (3, 64, 64), "circular"),
                                                     (2, (1, 17, 20), "reflect"), (3, (2, 18, 18), "replicate")])
def test_downsampling(dev, filt, factor, img_size, padding):
    import deepinv_amd as dinv

    g = _g(3)
    x = torch.rand(2, *img_size, generator=g)
    phys = dinv.physics.Downsampling(img_size=img_size, filter=filt, factor=factor, padding=padding, device=dev)
    k = None if filt is None else phys.filter.cpu()
    y = phys.A(x.to(dev))
    y_ref = O.downsampling_A(x, k, factor, padding)
    assert y.shape == y_ref.shape
    assert rel_err(y, y_ref) < TOL
    v = torch.randn(y_ref.shape, generator=g)
    assert rel_err(phys.A_adjoint(v.to(dev)), O.downsampling_AT(v, k, factor, img_size, padding)) < TOL
    assert dot_test(phys, x.to(dev), y) < 1e-5
    if padding == "circular" and filt is not None and img_size[1] % factor == 0 and img_size[2] % factor == 0:
        z = torch.rand(2, *img_size, generator=g)
        p = phys.prox_l2(z.to(dev), y, 0.8)
        assert rel_err(p, O.downsampling_prox_l2(z, y_ref, 0.8, k, factor, img_size)) < TOL


def test_rfft2_roundtrip_and_autograd(dev):
    from deepinv_amd.hip import conv as hc

    x = torch.randn(2, 3, 12, 10, generator=_g(4)).to(dev).requires_grad_(True)
    X = hc.rfft2(x, norm="ortho")
    assert rel_err(torch.view_as_real(X), torch.view_as_real(torch.fft.rfft2(x.detach().cpu(), norm="ortho"))) < TOL
    assert rel_err(hc.irfft2(X, (12, 10), norm="ortho"), x) < TOL
    w = torch.randn(2, 3, 12, 6, 2, generator=_g(5)).to(dev)
    (torch.view_as_real(X) * w).sum().backward()
    xc = x.detach().cpu().requires_grad_(True)
    (torch.view_as_real(torch.fft.rfft2(xc, norm="ortho")) * w.cpu()).sum().backward()
    assert rel_err(x.grad, xc.grad) < TOL
