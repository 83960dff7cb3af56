"""csrc/blur.hip on the host emulation (tests/emu): padded true convolution with stride (Blur / Downsampling.A), its exact
transpose (dot test), and the real <-> half-complex 2-D FFT of BlurFFT, against fp64 references - CPU-only coverage of the
kernel sources (deepinv/physics/functional/convolution.py:42-164, 689-758; deepinv/physics/blur.py:255-329, 639-657)."""
import ctypes

import numpy as np
import pytest
import torch

import emu_lib as E


class ConvDesc(ctypes.Structure):
    _fields_ = [("batch", ctypes.c_int32), ("channels", ctypes.c_int32), ("height", ctypes.c_int32), ("width", ctypes.c_int32),
                ("fbatch", ctypes.c_int32), ("fchannels", ctypes.c_int32), ("fh", ctypes.c_int32), ("fw", ctypes.c_int32),
                ("mode", ctypes.c_int32), ("stride", ctypes.c_int32)]


MODES = {"valid": 0, "circular": 1, "reflect": 2, "replicate": 3, "constant": 4}


def _ref_conv(x, k, mode, stride):
    """true convolution (flipped filter) of the padded image, deepinv's `conv2d` (convolution.py:42-107), then [::s, ::s]"""
    B, C, H, W = x.shape
    fh, fw = k.shape[-2:]
    x, k = x.double(), k.double().expand(-1, C, -1, -1) if k.shape[1] != C else k.double()
    if mode != "valid":
        ph, pw = fh // 2, fw // 2
        pad = (pw, pw - (1 - fw % 2), ph, ph - (1 - fh % 2))       # deepinv pads ih = (h-1)//2 .. : the odd / even split below
        pad = ((fw - 1) // 2, fw // 2, (fh - 1) // 2, fh // 2)
        x = torch.nn.functional.pad(x, pad, mode=mode if mode != "constant" else "constant", value=0)
    out = []
    for b in range(B):
        kb = k[b if k.shape[0] > 1 else 0]
        out.append(torch.nn.functional.conv2d(x[b:b + 1], torch.flip(kb, (-2, -1))[:, None], groups=C))
    y = torch.cat(out, 0)
    return y[:, :, ::stride, ::stride]


@pytest.mark.parametrize("mode", ["valid", "circular", "reflect", "replicate", "constant"])
@pytest.mark.parametrize("shape", [(2, 3, 12, 15, 1, 1, 3, 3, 1), (1, 2, 16, 16, 1, 2, 5, 5, 2), (2, 1, 20, 12, 2, 1, 4, 4, 4),
                                   (1, 3, 9, 11, 1, 3, 3, 5, 1),
                                   # the strided LDS-tiled kernels: several tiles, ragged edges, filters wider than the stride (bicubic x4 =
                                   # 16 x 16 at stride 4), sizes that the stride does not divide (circular transposed: the general kernel)
                                   (1, 2, 72, 80, 1, 1, 16, 16, 4), (2, 1, 70, 66, 2, 1, 8, 6, 2), (1, 1, 45, 70, 1, 1, 7, 9, 3),
                                   (1, 1, 132, 40, 1, 1, 5, 5, 4)])
def test_conv2d_and_transpose_emulated(mode, shape):
    B, C, H, W, fb, fc, fh, fw, s = shape
    gen = torch.Generator().manual_seed(H * W + fh)
    x = torch.randn(B, C, H, W, generator=gen)
    k = torch.randn(fb, fc, fh, fw, generator=gen)
    l = E.lib()
    d = ConvDesc(B, C, H, W, fb, fc, fh, fw, MODES[mode], s)
    ho, wo = ctypes.c_int32(), ctypes.c_int32()
    E.check(l.dinv_conv2d_out_size(ctypes.byref(d), ctypes.byref(ho), ctypes.byref(wo)))
    y = torch.full((B, C, ho.value, wo.value), float("nan"))
    E.check(l.dinv_conv2d(ctypes.byref(d), E.p(x), E.p(k), E.p(y), None))
    ref = _ref_conv(x, k, mode, s)
    assert tuple(ref.shape) == tuple(y.shape), (ref.shape, y.shape)
    assert float((y.double() - ref).norm() / ref.norm()) < 2e-6
    # exact transpose: <A x, v> = <x, A^T v>
    v = torch.randn(*y.shape, generator=gen)
    xt = torch.full((B, C, H, W), float("nan"))
    E.check(l.dinv_conv2d_transpose(ctypes.byref(d), E.p(v), E.p(k), E.p(xt), None))
    lhs, rhs = float((y.double() * v.double()).sum()), float((x.double() * xt.double()).sum())
    assert abs(lhs - rhs) < 1e-5 * max(abs(lhs), 1.0, 0.01 * float(y.double().norm() * v.double().norm()))   # (|<Ax, v>| can be small by cancellation)
    xd = x.double().requires_grad_(True)
    (_ref_conv(xd, k, mode, s) * v.double()).sum().backward()
    assert float((xt.double() - xd.grad).norm() / xd.grad.norm()) < 2e-6        # and the adjoint itself against autograd in fp64
    # gradient w.r.t. the filter, per (b, c) plane, against autograd through the fp64 reference convolution with a per-plane filter
    kp = k.expand(B, C, fh, fw).clone().double().requires_grad_(True)
    (_ref_conv(x, kp, mode, s) * v.double()).sum().backward()
    dk = torch.full((B, C, fh, fw), float("nan"))
    E.check(l.dinv_conv2d_filter_grad(ctypes.byref(d), E.p(x), E.p(v), E.p(dk), None))
    assert float((dk.double() - kp.grad).norm() / kp.grad.norm()) < 2e-6


@pytest.mark.parametrize("H,W", [(16, 32), (12, 20), (32, 64), (9, 14)])
def test_rfft2_irfft2_emulated(H, W):
    P = 3
    gen = torch.Generator().manual_seed(H + W)
    x = torch.randn(P, H, W, generator=gen)
    l = E.lib()
    ph, th = E.fft_plan(H)
    pw, tw = E.fft_plan(W)
    out = torch.full((P, H, W // 2 + 1, 2), float("nan"))
    E.check(l.dinv_rfft2(E.p(x), E.p(out), ctypes.c_int64(P), ctypes.byref(ph), E.p(th), ctypes.byref(pw), E.p(tw),
                         ctypes.c_float(1.0), None))
    ref = torch.view_as_real(torch.fft.rfft2(x.double()))
    assert float((out.double() - ref).norm() / ref.norm()) < 2e-6
    back = torch.full((P, H, W), float("nan"))
    ws = np.zeros(P * H * (W // 2 + 1) * 8, np.uint8)
    E.check(l.dinv_irfft2(E.p(out), E.p(back), ctypes.c_int64(P), ctypes.byref(ph), E.p(th), ctypes.byref(pw), E.p(tw),
                          ctypes.c_float(1.0 / (H * W)), E.p(ws), ctypes.c_size_t(ws.size), None))
    assert float((back.double() - x.double()).norm() / x.double().norm()) < 2e-6


class Conv3dDesc(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in ("batch", "channels", "depth", "height", "width", "fbatch", "fchannels", "fd", "fh", "fw",
                                              "mode", "reserved")]


def _ref_conv3(x, k, mode):
    """true 3-D convolution of the padded volume, deepinv's conv3d (convolution.py:333-393)"""
    B, C = x.shape[:2]
    fd, fh, fw = k.shape[-3:]
    if mode != "valid":
        pad = ((fw - 1) // 2, fw // 2, (fh - 1) // 2, fh // 2, (fd - 1) // 2, fd // 2)
        x = torch.nn.functional.pad(x, pad, mode=mode if mode != "constant" else "constant", value=0)
    k = k.expand(B, C, fd, fh, fw)
    out = torch.nn.functional.conv3d(x.reshape(1, B * C, *x.shape[2:]), torch.flip(k, (-3, -2, -1)).reshape(B * C, 1, fd, fh, fw), groups=B * C)
    return out.reshape(B, C, *out.shape[2:])


@pytest.mark.parametrize("mode", ["valid", "circular", "reflect", "replicate", "constant"])
@pytest.mark.parametrize("shape", [(2, 2, 6, 9, 11, 1, 1, 3, 3, 3), (1, 3, 7, 8, 10, 1, 3, 2, 4, 3), (2, 1, 5, 12, 9, 2, 1, 3, 5, 2)])
def test_conv3d_transpose_and_filter_grad_emulated(mode, shape):
    """the volume kernels of csrc/blur.hip (conv3d / conv_transpose3d, convolution.py:333-452): against an fp64 grouped conv3d of
    the padded volume, the dot test, and autograd of the reference form for the filter gradient; even and odd filter sizes"""
    B, C, D, H, W, fb, fc, fd, fh, fw = shape
    gen = torch.Generator().manual_seed(D * H * W + fd)
    x = torch.randn(B, C, D, H, W, generator=gen)
    k = torch.randn(fb, fc, fd, fh, fw, generator=gen)
    l = E.lib()
    d = Conv3dDesc(B, C, D, H, W, fb, fc, fd, fh, fw, MODES[mode], 0)
    do, ho, wo = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
    E.check(l.dinv_conv3d_out_size(ctypes.byref(d), ctypes.byref(do), ctypes.byref(ho), ctypes.byref(wo)))
    y = torch.full((B, C, do.value, ho.value, wo.value), float("nan"))
    E.check(l.dinv_conv3d(ctypes.byref(d), E.p(x), E.p(k), E.p(y), None))
    kp = k.expand(B, C, fd, fh, fw).clone().double().requires_grad_(True)
    ref = _ref_conv3(x.double(), kp, mode)
    assert tuple(ref.shape) == tuple(y.shape)
    assert float((y.double() - ref.detach()).norm() / ref.detach().norm()) < 2e-6
    v = torch.randn(*y.shape, generator=gen)
    xt = torch.full((B, C, D, H, W), float("nan"))
    E.check(l.dinv_conv3d_transpose(ctypes.byref(d), E.p(v), E.p(k), E.p(xt), None))
    lhs, rhs = float((y.double() * v.double()).sum()), float((x.double() * xt.double()).sum())
    assert abs(lhs - rhs) < 1e-5 * max(abs(lhs), 1.0)
    (ref * v.double()).sum().backward()
    dk = torch.full((B, C, fd, fh, fw), float("nan"))
    E.check(l.dinv_conv3d_filter_grad(ctypes.byref(d), E.p(x), E.p(v), E.p(dk), None))
    assert float((dk.double() - kp.grad).norm() / kp.grad.norm()) < 2e-6


def _symbol_ref(X, m, a, flags, add):
    """SYMBOL of include/deepinv_amd.h (dinv_blurfft_apply) in fp64 with the reference's expressions (blur.py:639-657,
    forward.py:1080-1117, 1212-1252): X complex [P,H,Wh], m real pairs [Ps,H,Wh,2], a complex [Ps,H,Wh]"""
    P, Ps = X.shape[0], m.shape[0]
    m = m.double().repeat(P // Ps, 1, 1, 1)
    a = a.to(torch.complex128).repeat(P // Ps, 1, 1)
    v = X.to(torch.complex128)
    if flags & 1:
        v = v * torch.conj(a)
    v = torch.view_as_real(v)
    mode = (flags >> 4) & 7
    if mode == 1:
        v = m * v
    elif mode == 2:
        v = m * m * v
    elif mode == 3:
        v = v / (m * m + add)
    elif mode == 4:
        v = v * torch.where(m > 1e-5, 1 / m, torch.zeros_like(m))
    v = torch.view_as_complex(v.contiguous())
    if flags & 2:
        v = v * a
    return v


BLURFFT_FLAGS = [0x10 | 2, 1 | 0x10, 0x20, 1 | 0x20 | 2, 0x30, 1 | 0x40, 0]


@pytest.mark.parametrize("H,W,Ps", [(64, 32, 3), (64, 64, 1), (16, 24, 3), (12, 10, 6), (128, 20, 2)])
def test_blurfft_apply_emulated(H, W, Ps):
    """dinv_blurfft_apply (csrc/blur.hip): out = irfft2(SYMBOL(rfft2 x)) for every operator of BlurFFT / DecomposablePhysics, against
    an fp64 evaluation of the reference's expressions - the fused LDS column pass (H = 64, 128), the three-pass form of other
    heights, widths that are / are not multiples of 4, symbols shared by the batch (Ps = C) and per sample (Ps = B C); and
    dinv_spectrum_symbol alone"""
    P, Wh = 6, W // 2 + 1
    gen = torch.Generator().manual_seed(H * W + Ps)
    x = torch.randn(P, H, W, generator=gen)
    m = torch.rand(Ps, H, Wh, 2, generator=gen) + 0.05
    m[:, 0, 0] = 1e-7                                   # a singular value below the pseudo-inverse's threshold
    ph_ = torch.rand(Ps, H, Wh, generator=gen) * 6.28
    a = torch.polar(torch.ones_like(ph_), ph_).contiguous()
    l = E.lib()
    l.dinv_blurfft_workspace_bytes.restype = ctypes.c_size_t
    ph, th = E.fft_plan(H)
    pw, tw = E.fft_plan(W)
    nb = l.dinv_blurfft_workspace_bytes(ctypes.c_int64(P), ctypes.c_int32(H), ctypes.c_int32(W))
    X = torch.fft.rfft2(x.double(), norm="ortho")
    for flags in BLURFFT_FLAGS:
        add = 1.0 / 1.3
        ws = torch.zeros(nb // 4 + 4)
        out = torch.full((P, H, W), float("nan"))
        E.check(l.dinv_blurfft_apply(E.p(x), E.p(out), ctypes.c_int64(P), ctypes.byref(ph), E.p(th), ctypes.byref(pw), E.p(tw), E.p(m),
                                     E.p(torch.view_as_real(a)), ctypes.c_int64(Ps), ctypes.c_int32(flags), ctypes.c_float(add),
                                     ctypes.c_float(1.0 / (H * W)), E.p(ws), ctypes.c_size_t(ws.numel() * 4), None))
        S = _symbol_ref(X, m, a, flags, add)
        ref = torch.fft.irfft2(S, s=(H, W), norm="ortho")
        err = float((out.double() - ref).norm() / ref.norm())
        assert err < 3e-6, (flags, err)
        spec = X.to(torch.complex64).contiguous()
        so = torch.full((P, H, Wh, 2), float("nan"))
        E.check(l.dinv_spectrum_symbol(E.p(torch.view_as_real(spec)), E.p(so), ctypes.c_int64(P), ctypes.c_int32(H), ctypes.c_int32(Wh),
                                       E.p(m), E.p(torch.view_as_real(a)), ctypes.c_int64(Ps), ctypes.c_int32(flags),
                                       ctypes.c_float(add), None))
        assert float((torch.view_as_complex(so).to(torch.complex128) - S).norm() / S.norm()) < 1e-6, flags
    # a symbol that needs buffers it was not given is an error, not a crash
    assert l.dinv_blurfft_apply(E.p(x), E.p(out), ctypes.c_int64(P), ctypes.byref(ph), E.p(th), ctypes.byref(pw), E.p(tw), None, None,
                                ctypes.c_int64(Ps), ctypes.c_int32(0x10), ctypes.c_float(0), ctypes.c_float(1), E.p(ws),
                                ctypes.c_size_t(ws.numel() * 4), None) != 0


def test_blurfft_class_on_the_fused_operator_emulated():
    """the product's BlurFFT class over the emulated kernels: A / A_adjoint / A_adjoint_A / A_A_adjoint / prox_l2 / A_dagger take the
    fused call and equal the oracle's restatement of the reference (oracle/physics_cpu.py); U / U_adjoint multiply through
    dinv_spectrum_symbol"""
    import deepinv_amd as dinv
    from emu_backend import emu_backend
    from oracle import physics_cpu as O

    img = (3, 64, 32)
    g = torch.Generator().manual_seed(11)
    x = torch.rand(2, *img, generator=g)
    k = torch.rand(1, 1, 5, 5, generator=g)
    k = k / k.sum()
    mask, angle = O.blurfft_params(img, k)
    with emu_backend():
        from deepinv_amd.hip import conv as hc
        calls = []
        orig = hc.blurfft_apply
        hc.blurfft_apply = lambda *a, **kw: (calls.append(a[3]), orig(*a, **kw))[1]
        try:
            p = dinv.physics.BlurFFT(img_size=img, filter=k, device="cpu")
            y = p.A(x)
            yr = O.blurfft_A(x, mask, angle, img)
            assert float((y - yr).norm() / yr.norm()) < 1e-5
            assert float((p.A_adjoint(y) - O.blurfft_AT(yr, mask, angle, img)).norm() / yr.norm()) < 1e-5
            z = torch.rand(2, *img, generator=g)
            pr = O.blurfft_prox_l2(z, yr, 1.3, mask, angle, img)
            assert float((p.prox_l2(z, y, 1.3) - pr).norm() / pr.norm()) < 1e-5
            ata = O.blurfft_AT(O.blurfft_A(x, mask, angle, img), mask, angle, img)
            assert float((p.A_adjoint_A(x) - ata).norm() / ata.norm()) < 1e-5
            aat = O.blurfft_A(O.blurfft_AT(yr, mask, angle, img), mask, angle, img)
            assert float((p.A_A_adjoint(y) - aat).norm() / aat.norm()) < 1e-5
            dag = p.A_dagger(y)
            assert float((p.A(dag) - y).norm() / y.norm()) < 1e-3          # A A^+ y = y on the range
            assert len(calls) >= 8 and {0x12, 0x11, 0x20, 0x23, 0x30, 0x41} <= set(calls), calls
            u = p.U(p.V_adjoint(x))
            ur = torch.fft.irfft2(torch.fft.rfft2(x, norm="ortho") * angle, s=img[-2:], norm="ortho")
            assert float((u - ur).norm() / ur.norm()) < 1e-5
            ua = torch.view_as_complex(p.U_adjoint(x).contiguous())
            uar = torch.fft.rfft2(x, norm="ortho") * torch.conj(angle)
            assert float((ua - uar).norm() / uar.norm()) < 1e-5
        finally:
            hc.blurfft_apply = orig
