"""The bf16-split DRUNet convolution kernels (deepinv_amd/csrc/drunet_split2d.hip, drunet_bf16s.hip) executed on the HOST by the fiber
emulation of tests/emu (MFMA emulated from the documented fragment layout) against an fp64 convolution."""
import ctypes

import numpy as np
import pytest
import torch

import emu_lib as E


class ActGeom(ctypes.Structure):
    _fields_ = [("batch", ctypes.c_int32), ("height", ctypes.c_int32), ("width", ctypes.c_int32), ("hp", ctypes.c_int32),
                ("wp", ctypes.c_int32), ("plane", ctypes.c_int64), ("np", ctypes.c_int64), ("sl", ctypes.c_int64),
                ("cs", ctypes.c_int64)]


def geom(B, H, W):
    """dinv_act_geom_init (csrc/drunet.hip) restated"""
    g = ActGeom()
    g.batch, g.height, g.width = B, H, W
    g.hp, g.wp = H + 2, (W + 2 + 3) // 4 * 4
    g.plane = g.hp * g.wp
    g.np = g.plane * B
    g.sl = g.wp + 4
    g.cs = g.sl + max((g.np + 511) // 512 * 512 + g.wp + 4, g.np + 34 * g.wp + 64)
    g.cs = (g.cs + 3) // 4 * 4
    return g


def to_act(t, g):
    B, C, H, W = t.shape
    if C % 8:       # channel blocks are 8 wide: pad with zero channels
        t = torch.cat((t, torch.zeros(B, 8 - C % 8, H, W, dtype=t.dtype)), 1)
        C = t.shape[1]
    a = torch.zeros(C // 8, g.cs, 8)
    av = a[:, g.sl:g.sl + g.np].view(-1, B, g.hp, g.wp, 8)
    av[:, :, 1:H + 1, 1:W + 1] = t.view(B, -1, 8, H, W).permute(1, 0, 3, 4, 2)
    return a


def from_act(a, g, C):
    B, H, W = g.batch, g.height, g.width
    av = a[:, g.sl:g.sl + g.np].view(-1, B, g.hp, g.wp, 8)
    return av[:, :, 1:H + 1, 1:W + 1].permute(1, 0, 4, 2, 3).reshape(B, C, H, W)


def pack(w):
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from deepinv_amd.hip.drunet import pack_split2d_weight   # pure torch host code
    return pack_split2d_weight(w)


def split_to_act(t, g):
    """pre-split activation buffer: per pixel and 8-channel block, 8 bf16 high parts then 8 bf16 low parts (32 bytes)"""
    a = to_act(t, g)
    hi = a.bfloat16()
    lo = (a - hi.float()).bfloat16()
    return torch.cat((hi, lo), dim=-1).view(torch.float32).contiguous()      # [C/8, cs, 8] fp32 words holding 16 bf16


def split_from_act(a, g, C):
    h = a.view(torch.bfloat16).view(a.shape[0], a.shape[1], 16)
    return from_act((h[..., :8].float() + h[..., 8:].float()).contiguous(), g, C)


@pytest.mark.parametrize("B,H,W,cin,cout,mode", [(1, 12, 20, 16, 64, "plain"), (2, 9, 14, 32, 64, "relu"),
                                                 (1, 17, 33, 32, 128, "res"), (3, 16, 16, 64, 64, "plain"),
                                                 (1, 8, 32, 16, 64, "relu"), (2, 40, 48, 16, 128, "res"),
                                                 (2, 40, 64, 32, 64, "res"), (2, 9, 14, 32, 64, "gate"),
                                                 (1, 17, 33, 16, 128, "gate")])
@pytest.mark.parametrize("fmt", ["f32", "in_split", "out_split"])
@pytest.mark.parametrize("tile", [1, 2])     # 128- / 256-pixel workgroups
def test_split2d_conv_matches_fp64(B, H, W, cin, cout, mode, fmt, tile):
    """csrc/drunet_split2d.hip (2-D tiles, optional pre-split activations) against an fp64 convolution; tile widths 32 /
    16 / 8 incl. partial column tiles (W = 20, 14, 33) and row tiles that straddle images"""
    if fmt == "out_split" and mode == "res":
        pytest.skip("a pre-split output carries no residual")
    if mode == "gate" and fmt != "f32":
        pytest.skip("the gate (ReLU backward in the epilogue) is an fp32 in / out mode")
    gen = torch.Generator().manual_seed(H * W + cin)
    x = torch.randn(B, cin, H, W, generator=gen)
    w = torch.randn(cout, cin, 3, 3, generator=gen) / (3.0 * cin ** 0.5)
    r = torch.randn(B, cout, H, W, generator=gen)
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from deepinv_amd.hip.drunet import pack_split2d_weight
    g = geom(B, H, W)
    if fmt == "in_split":
        xa = split_to_act(x, g)
        x = split_from_act(xa, g, cin)              # the operand the kernel really sees
    else:
        xa = to_act(x, g)
    ref = torch.nn.functional.conv2d(x.double(), w.double(), padding=1)
    if mode == "relu":
        ref = ref.relu()
    if mode == "res":
        ref = ref + r.double()
    if mode == "gate":        # y = r > 0 ? conv : 0  (r = the ReLU output of the forward pass)
        r = r.relu()
        ref = ref * (r > 0)
    ra = to_act(r, g)
    ya = torch.full((cout // 8, g.cs, 8), float("nan"))
    ya[:, :g.sl] = 0
    ya[:, g.sl + g.np:] = 0
    ya[:, g.sl:g.sl + g.np].view(-1, B, g.hp, g.wp, 8)[:, :, :, 0] = 0        # frame columns are never written: must stay zero
    ya[:, g.sl:g.sl + g.np].view(-1, B, g.hp, g.wp, 8)[:, :, :, W + 1:] = 0
    wp = pack_split2d_weight(w)
    flags = (1 if fmt == "in_split" else 0) | (2 if fmt == "out_split" else 0) | (4 if mode == "relu" else 0) | (tile << 8)
    flags |= 8 if mode == "gate" else 0
    l = E.lib()
    E.check(l.dinv_conv3x3_split(ctypes.byref(g), E.p(xa), ctypes.c_void_p(wp.data_ptr()), cin, cout, E.p(ya),
                                 E.p(ra) if mode in ("res", "gate") else None, flags, None))
    assert not torch.isnan(ya).any()
    out = split_from_act(ya, g, cout) if fmt == "out_split" else from_act(ya, g, cout)
    err = float((out.double() - ref).norm() / ref.norm())
    assert err < 2e-5, err
    # the zero frame survives (rows between images are written as zeros, frame columns untouched)
    full = ya[:, g.sl:g.sl + g.np].view(-1, B, g.hp, g.wp, 8)
    assert float(full[:, :, 0].abs().max()) == 0 and float(full[:, :, H + 1].abs().max()) == 0
    assert float(full[:, :, :, 0].abs().max()) == 0 and float(full[:, :, :, W + 1:].abs().max()) == 0


@pytest.mark.parametrize("B,H,W,cin,cout,mode", [(1, 12, 20, 16, 64, "plain"), (2, 9, 14, 32, 64, "relu"),
                                                 (1, 17, 34, 32, 128, "res"), (3, 16, 16, 64, 64, "plain"),
                                                 (1, 8, 32, 16, 64, "relu"), (2, 40, 48, 16, 128, "res"),
                                                 (2, 40, 64, 32, 64, "res"), (1, 36, 8, 48, 64, "relu"), (1, 5, 6, 16, 64, "plain")])
def test_wsplit_conv_matches_fp64(B, H, W, cin, cout, mode):
    """csrc/drunet_wsplit.hip (Winograd F(2,3) along rows, bf16-split operands, wave = Winograd point, cross-wave exchange in
    the epilogue) against an fp64 convolution; tile widths 32 / 16 / 8 incl. partial column tiles (W = 20, 14, 34, 6), row
    tiles that straddle images, odd step counts (cin = 16, 48)"""
    gen = torch.Generator().manual_seed(H * W + cin)
    x = torch.randn(B, cin, H, W, generator=gen)
    w = torch.randn(cout, cin, 3, 3, generator=gen) / (3.0 * cin ** 0.5)
    r = torch.randn(B, cout, H, W, generator=gen)
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from deepinv_amd.hip.drunet import pack_wsplit_weight
    g = geom(B, H, W)
    xa = to_act(x, g)
    ref = torch.nn.functional.conv2d(x.double(), w.double(), padding=1)
    if mode == "relu":
        ref = ref.relu()
    if mode == "res":
        ref = ref + r.double()
    ra = to_act(r, g)
    ya = torch.full((cout // 8, g.cs, 8), float("nan"))
    ya[:, :g.sl] = 0
    ya[:, g.sl + g.np:] = 0
    ya[:, g.sl:g.sl + g.np].view(-1, B, g.hp, g.wp, 8)[:, :, :, 0] = 0        # frame columns are never written: must stay zero
    ya[:, g.sl:g.sl + g.np].view(-1, B, g.hp, g.wp, 8)[:, :, :, W + 1:] = 0
    wp = pack_wsplit_weight(w)
    l = E.lib()
    E.check(l.dinv_conv3x3_wsplit(ctypes.byref(g), E.p(xa), ctypes.c_void_p(wp.data_ptr()), cin, cout, E.p(ya),
                                  E.p(ra) if mode == "res" else None, 4 if mode == "relu" else 0, None))
    assert not torch.isnan(ya).any()
    out = from_act(ya, g, cout)
    err = float((out.double() - ref).norm() / ref.norm())
    assert err < 2e-5, err
    full = ya[:, g.sl:g.sl + g.np].view(-1, B, g.hp, g.wp, 8)
    assert float(full[:, :, 0].abs().max()) == 0 and float(full[:, :, H + 1].abs().max()) == 0
    assert float(full[:, :, :, 0].abs().max()) == 0 and float(full[:, :, :, W + 1:].abs().max()) == 0


def test_wsplit_rejects_odd_width():
    g = geom(1, 8, 9)
    l = E.lib()
    x = torch.zeros((2, g.cs, 8))
    y = torch.zeros((8, g.cs, 8))
    w = torch.zeros(64 * 16 * 9 * 4, dtype=torch.bfloat16)
    assert l.dinv_conv3x3_wsplit(ctypes.byref(g), E.p(x), ctypes.c_void_p(w.data_ptr()), 16, 64, E.p(y), None, 0, None) != 0


def wide_range(shape, lo_exp, hi_exp, gen):
    """random signs and mantissas, exponents uniform in [lo_exp, hi_exp]: operands spanning many binades"""
    e = torch.randint(lo_exp, hi_exp + 1, shape, generator=gen).float()
    return (1 + torch.rand(shape, generator=gen)) * torch.exp2(e) * (torch.randint(0, 2, shape, generator=gen) * 2 - 1).float()


@pytest.mark.parametrize("case", ["wide", "he_scale"])
def test_split_worst_case(case):
    """The operand split's WORST-CASE bound, checked element by element: x = xh + xl + e with |e| <= 2^-16 |x| (xl is the
    bf16 rounding of an exact remainder of at most 2^-8 |x|), the dropped al*bl product is at most 2^-16 |a||b|, so every
    output obeys |y - y_exact| <= 3 * 2^-16 * (|w| conv |x|) + fp32 accumulation (bounded here by 2^-20 of the same sum).
    `wide`: activations spanning 2^-20 .. 2^8 and weights 2^-12 .. 2^2 (no subnormal low parts: bf16 keeps the fp32
    exponent range); `he_scale`: O(1)-gain weights (std sqrt(2 / fan_in)) like a trained ResBlock, N(0,1) data, where the
    typical error must stay at the few-1e-6 level."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from deepinv_amd.hip.drunet import pack_split2d_weight
    gen = torch.Generator().manual_seed(11)
    B, H, W, cin, cout = 2, 16, 32, 64, 64
    if case == "wide":
        x, w = wide_range((B, cin, H, W), -20, 8, gen), wide_range((cout, cin, 3, 3), -12, 2, gen)
    else:
        x, w = torch.randn(B, cin, H, W, generator=gen), torch.randn(cout, cin, 3, 3, generator=gen) * (2.0 / (9 * cin)) ** 0.5
    ref = torch.nn.functional.conv2d(x.double(), w.double(), padding=1)
    mag = torch.nn.functional.conv2d(x.double().abs(), w.double().abs(), padding=1)
    g = geom(B, H, W)
    xa = to_act(x, g)
    ya = torch.zeros(cout // 8, g.cs, 8)
    l = E.lib()
    wp = pack_split2d_weight(w)        # keep the packed tensor alive across the call
    E.check(l.dinv_conv3x3_split(ctypes.byref(g), E.p(xa), ctypes.c_void_p(wp.data_ptr()), cin, cout, E.p(ya), None, 0, None))
    out = from_act(ya, g, cout).double()
    bound = (3 * 2.0 ** -16 + 2.0 ** -20) * mag
    assert bool(((out - ref).abs() <= bound).all()), float(((out - ref).abs() / mag).max())
    # the bound is not vacuous: the observed worst element uses a visible part of it, the typical one far less
    worst = float(((out - ref).abs() / mag).max())
    assert worst < 3 * 2.0 ** -16
    if case == "he_scale":
        assert float((out - ref).norm() / ref.norm()) < 5e-6


def winograd_magnitude(x, w):
    """sum over the three Winograd points of an output pixel of |U_k| conv |V_k| (fp64): the quantity the operand-split error
    of csrc/drunet_wsplit.hip is relative to.  Even columns use points 0, 1, 2, odd columns 1, 2, 3."""
    B, C, H, W = x.shape
    g = w.double()
    U = torch.stack((g[..., 0], (g[..., 0] + g[..., 1] + g[..., 2]) / 2, (g[..., 0] - g[..., 1] + g[..., 2]) / 2, g[..., 2])).abs()
    xp = torch.nn.functional.pad(x.double(), (1, 1, 1, 1))
    d0, d1, d2, d3 = xp[..., 0:W:2], xp[..., 1:W + 1:2], xp[..., 2:W + 2:2], xp[..., 3:W + 3:2]
    V = [(d0 - d2).abs(), (d1 + d2).abs(), (d2 - d1).abs(), (d1 - d3).abs()]
    M = [torch.nn.functional.conv2d(V[k], U[k].unsqueeze(-1)) for k in range(4)]         # rows only: kernel [Co, Ci, 3, 1]
    return torch.stack((M[0] + M[1] + M[2], M[1] + M[2] + M[3]), -1).reshape(B, -1, H, W)


@pytest.mark.parametrize("case", ["wide", "he_scale"])
def test_wsplit_worst_case(case):
    """The worst-case bound of the Winograd operand-split kernel, element by element.  Every multiply U_k V_k is evaluated as
    Uh Vl + Ul Vh + Uh Vh with U, V = high part + low part + (at most 2^-16 of themselves), so each output obeys
    |y - y_exact| <= 3 * 2^-16 * sum_k (|U_k| conv |V_k|) + the fp32 roundings of V, of the accumulation and of the output
    transform (bounded here by 2^-20 of the same sum) - the bound of the direct kernel with |U| (x) |V| in the place of
    |w| (x) |x|.  `wide`: activations spanning 2^-20 .. 2^8, weights 2^-12 .. 2^2; `he_scale`: O(1)-gain weights and N(0,1)
    data, where the typical error must stay at the few-1e-6 level."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from deepinv_amd.hip.drunet import pack_wsplit_weight
    gen = torch.Generator().manual_seed(12)
    B, H, W, cin, cout = 2, 16, 32, 64, 64
    if case == "wide":
        x, w = wide_range((B, cin, H, W), -20, 8, gen), wide_range((cout, cin, 3, 3), -12, 2, gen)
    else:
        x, w = torch.randn(B, cin, H, W, generator=gen), torch.randn(cout, cin, 3, 3, generator=gen) * (2.0 / (9 * cin)) ** 0.5
    ref = torch.nn.functional.conv2d(x.double(), w.double(), padding=1)
    mag = winograd_magnitude(x, w)
    g = geom(B, H, W)
    xa = to_act(x, g)
    ya = torch.zeros(cout // 8, g.cs, 8)
    l = E.lib()
    wp = pack_wsplit_weight(w)
    E.check(l.dinv_conv3x3_wsplit(ctypes.byref(g), E.p(xa), ctypes.c_void_p(wp.data_ptr()), cin, cout, E.p(ya), None, 0, None))
    out = from_act(ya, g, cout).double()
    bound = (3 * 2.0 ** -16 + 2.0 ** -20) * mag
    assert bool(((out - ref).abs() <= bound).all()), float(((out - ref).abs() / mag).max())
    assert float(((out - ref).abs() / mag).max()) < 3 * 2.0 ** -16
    if case == "he_scale":
        assert float((out - ref).norm() / ref.norm()) < 8e-6


@pytest.mark.parametrize("B,H,W,cin,cout", [(1, 8, 12, 16, 64), (2, 10, 6, 32, 128), (3, 16, 16, 64, 64)])
def test_bf16_split_down2x2_matches_fp64(B, H, W, cin, cout):
    """2x2 stride-2 convolution of drunet_bf16s.hip (downsample_strideconv, drunet.py:524-552) on the host emulation"""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from deepinv_amd.hip.drunet import pack_down_bf16s_weight

    gen = torch.Generator().manual_seed(H * W + cin)
    x = torch.randn(B, cin, H, W, generator=gen)
    w = torch.randn(cout, cin, 2, 2, generator=gen) / (2.0 * cin ** 0.5)
    ref = torch.nn.functional.conv2d(x.double(), w.double(), stride=2)
    gi, go = geom(B, H, W), geom(B, H // 2, W // 2)
    xa = to_act(x, gi)
    ya = torch.full((cout // 8, go.cs, 8), float("nan"))
    ya[:, :go.sl] = 0
    wp = pack_down_bf16s_weight(w)
    l = E.lib()
    E.check(l.dinv_conv_down2x2_bf16s(ctypes.byref(gi), ctypes.byref(go), E.p(xa), ctypes.c_void_p(wp.data_ptr()), cin, cout,
                                      E.p(ya), None))
    out = from_act(ya, go, cout)
    assert not torch.isnan(out).any()
    err = float((out.double() - ref).norm() / ref.norm())
    assert err < 2e-5, err
    full = ya[:, go.sl:go.sl + go.np].view(-1, B, go.hp, go.wp, 8)       # the zero frame of the output is written as zeros
    assert float(full[:, :, 0].abs().max()) == 0 and float(full[:, :, :, 0].abs().max()) == 0


@pytest.mark.parametrize("B,H,W,cin,cout", [(1, 8, 12, 16, 64), (2, 10, 6, 32, 128), (3, 16, 16, 64, 64), (1, 4, 36, 128, 64)])
def test_bf16x3_down2x2_is_fp32_equivalent(B, H, W, cin, cout):
    """dinv_conv_down2x2_bf16x3: three-part operand split, six products (the stride-2 layers of conv_precision = "fp32").  Against an
    fp64 convolution it must sit at the level of fp32 arithmetic itself: measured 1.5e-7 ... 4.1e-7 where PyTorch's fp32 conv2d
    has 1.0e-7 ... 1.5e-7 on the same operands (operands carried to 24 bits, the three dropped cross terms are 2^-24 each, the
    accumulator is rounded once per 16-term MFMA and product) - the two-part kernel of the bf16split setting: 3e-6, the F(4x4)
    ResBlock kernel of the same setting: 3e-6.  Asserted: below 1e-6 and within 4x of the fp32 conv2d"""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from deepinv_amd.hip.drunet import pack_down_bf16x3_weight

    gen = torch.Generator().manual_seed(H * W + cin)
    x = torch.randn(B, cin, H, W, generator=gen)
    w = torch.randn(cout, cin, 2, 2, generator=gen) / (2.0 * cin ** 0.5)
    ref = torch.nn.functional.conv2d(x.double(), w.double(), stride=2)
    gi, go = geom(B, H, W), geom(B, H // 2, W // 2)
    xa = to_act(x, gi)
    ya = torch.full((cout // 8, go.cs, 8), float("nan"))
    ya[:, :go.sl] = 0
    wp = pack_down_bf16x3_weight(w)
    hi, mid, lo = (wp[:, :, :, k].float() for k in range(3))
    back = (hi.double() + mid.double() + lo.double()).permute(4, 2, 3, 5, 0, 1).reshape(cout, cin, 2, 2)    # co, (s, cblk, ci), dy, dx
    assert float((back - w.double()).abs().max() / w.abs().max()) < 2 ** -23          # the three planes carry the fp32 weight
    l = E.lib()
    E.check(l.dinv_conv_down2x2_bf16x3(ctypes.byref(gi), ctypes.byref(go), E.p(xa), ctypes.c_void_p(wp.data_ptr()), cin, cout,
                                       E.p(ya), None))
    out = from_act(ya, go, cout)
    assert not torch.isnan(out).any()
    err = float((out.double() - ref).norm() / ref.norm())
    err32 = float((torch.nn.functional.conv2d(x, w, stride=2).double() - ref).norm() / ref.norm())
    assert err < 1e-6 and err < 4 * err32 + 1e-7, (err, err32)
    full = ya[:, go.sl:go.sl + go.np].view(-1, B, go.hp, go.wp, 8)       # the zero frame of the output is written as zeros
    assert float(full[:, :, 0].abs().max()) == 0 and float(full[:, :, :, 0].abs().max()) == 0


@pytest.mark.parametrize("B,H,W,cin,cout,skip", [(1, 5, 7, 16, 64, False), (2, 6, 4, 32, 128, True), (3, 8, 8, 64, 64, True)])
def test_bf16_split_up2x2_matches_fp64(B, H, W, cin, cout, skip):
    """2x2 stride-2 transposed convolution of drunet_bf16s.hip (upsample_convtranspose, drunet.py:493-521), input = x (+ x2)"""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from deepinv_amd.hip.drunet import pack_up_bf16s_weight

    gen = torch.Generator().manual_seed(H * W + cin)
    x = torch.randn(B, cin, H, W, generator=gen)
    x2 = torch.randn(B, cin, H, W, generator=gen)
    w = torch.randn(cin, cout, 2, 2, generator=gen) / (cin ** 0.5)
    ref = torch.nn.functional.conv_transpose2d((x + x2 if skip else x).double(), w.double(), stride=2)
    gi, go = geom(B, H, W), geom(B, 2 * H, 2 * W)
    xa, x2a = to_act(x, gi), to_act(x2, gi)
    ya = torch.zeros((cout // 8, go.cs, 8))
    ya[:, go.sl:go.sl + go.np].view(-1, B, go.hp, go.wp, 8)[:, :, 1:2 * H + 1, 1:2 * W + 1] = float("nan")   # interior must be written
    wp = pack_up_bf16s_weight(w)
    l = E.lib()
    E.check(l.dinv_conv_up2x2_bf16s(ctypes.byref(gi), ctypes.byref(go), E.p(xa), E.p(x2a) if skip else None,
                                    ctypes.c_void_p(wp.data_ptr()), cin, cout, E.p(ya), None))
    out = from_act(ya, go, cout)
    assert not torch.isnan(ya).any()
    err = float((out.double() - ref).norm() / ref.norm())
    assert err < 2e-5, err
    full = ya[:, go.sl:go.sl + go.np].view(-1, B, go.hp, go.wp, 8)       # the zero frame is never touched
    assert float(full[:, :, 0].abs().max()) == 0 and float(full[:, :, :, 0].abs().max()) == 0
    assert float(full[:, :, 2 * H + 1].abs().max()) == 0 and float(full[:, :, :, 2 * W + 1:].abs().max()) == 0


def _wgrad(gs, gl, s, m, l, n, taps):
    lib = E.lib()
    lib.dinv_conv_wgrad_workspace_bytes.restype = ctypes.c_size_t
    k = 3 if taps == 9 else 2
    dw = torch.full((m, n, k, k), float("nan"))
    ws = torch.zeros(lib.dinv_conv_wgrad_workspace_bytes(ctypes.byref(gs), m, n, taps), dtype=torch.uint8)
    E.check(lib.dinv_conv_wgrad(ctypes.byref(gs), ctypes.byref(gl), E.p(s), m, E.p(l), n, taps, E.p(dw), 0, E.p(ws),
                                ctypes.c_size_t(ws.numel()), None))
    return dw


@pytest.mark.parametrize("B,H,W,cin,cout", [(2, 9, 14, 16, 64), (1, 12, 8, 3, 64), (3, 8, 8, 64, 2), (1, 20, 33, 40, 24),
                                            (2, 9, 14, 16, 16), (1, 7, 5, 3, 16), (2, 6, 6, 16, 2),     # thin-layer (16x16x4) kernel
                                            (2, 30, 29, 16, 16), (2, 30, 29, 24, 40)])                  # several workgroups per tile
def test_wgrad_3x3_matches_autograd(B, H, W, cin, cout):
    """dW of a 3x3 convolution (csrc/drunet_bwd.hip on the host emulation) vs torch autograd in fp64"""
    gen = torch.Generator().manual_seed(cin * cout + H)
    x = torch.randn(B, cin, H, W, generator=gen)
    gy = torch.randn(B, cout, H, W, generator=gen)
    w = torch.zeros(cout, cin, 3, 3, dtype=torch.float64, requires_grad=True)
    (torch.nn.functional.conv2d(x.double(), w, padding=1) * gy.double()).sum().backward()
    g = geom(B, H, W)
    dw = _wgrad(g, g, to_act(gy, g), cout, to_act(x, g), cin, 9)
    assert not torch.isnan(dw).any()
    assert float((dw.double() - w.grad).norm() / w.grad.norm()) < 1e-5


@pytest.mark.parametrize("B,H,W,cs,cl,up", [(2, 6, 8, 64, 16, False), (1, 5, 4, 32, 128, True), (2, 8, 8, 24, 40, False),
                                              (2, 6, 8, 16, 16, False), (1, 5, 4, 16, 8, True)])
def test_wgrad_2x2_matches_autograd(B, H, W, cs, cl, up):
    """dW of the 2x2 stride-2 convolution (S = dL/dy on the half grid) and of the transposed one (S = x on the half grid)"""
    gen = torch.Generator().manual_seed(cs + cl)
    small = torch.randn(B, cs, H, W, generator=gen)
    big = torch.randn(B, cl, 2 * H, 2 * W, generator=gen)
    w = torch.zeros(cs, cl, 2, 2, dtype=torch.float64, requires_grad=True)
    if up:      # y = convT(x = small, w [Cin, Cout, 2, 2]), dL/dy = big
        (torch.nn.functional.conv_transpose2d(small.double(), w, stride=2) * big.double()).sum().backward()
    else:       # y = conv(x = big, w [Cout, Cin, 2, 2], stride 2), dL/dy = small
        (torch.nn.functional.conv2d(big.double(), w, stride=2) * small.double()).sum().backward()
    gs, gl = geom(B, H, W), geom(B, 2 * H, 2 * W)
    dw = _wgrad(gs, gl, to_act(small, gs), cs, to_act(big, gl), cl, 4)
    assert not torch.isnan(dw).any()
    assert float((dw.double() - w.grad).norm() / w.grad.norm()) < 1e-5


def test_relu_backward():
    a = torch.randn(1000)
    g = torch.randn(1000)
    ref = g * (a > 0)
    E.check(E.lib().dinv_relu_backward(ctypes.c_int64(1000), E.p(a), E.p(g), None))
    assert torch.equal(g, ref)


# ---- 3-D volumes as stacks of slices (models/drunet3d.py): D + 2 images per volume, zero slices at both ends
def vol_to_act(t, g):
    """[B, C, D, H, W] -> activation buffer over B (D + 2) images"""
    B, C, D, H, W = t.shape
    t2 = torch.nn.functional.pad(t.permute(0, 2, 1, 3, 4), (0, 0, 0, 0, 0, 0, 1, 1)).reshape(B * (D + 2), C, H, W)
    return to_act(t2, g)


def act_to_vol(a, g, C, B, D):
    t2 = from_act(a, g, C)                           # [B (D + 2), C, H, W]
    return t2.view(B, D + 2, C, g.height, g.width)


@pytest.mark.parametrize("B,D,H,W,cin,cout", [(1, 2, 4, 6, 16, 64), (2, 4, 6, 4, 32, 64)])
def test_down_and_up_2x2x2_with_depth_pairing(B, D, H, W, cin, cout):
    """dinv_conv_down2x2_bf16s_3d / dinv_conv_up2x2_bf16s_3d (two depth taps, slice pairing z <-> 2 z + dz inside the
    kernel, zero slices kept zero) against conv3d / conv_transpose3d with 2x2x2 kernels and stride 2"""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from deepinv_amd.hip.drunet import pack_down_bf16s_weight, pack_up_bf16s_weight

    gen = torch.Generator().manual_seed(D * H + cin)
    l = E.lib()
    # ---- down: [B, cin, 2D, 2H, 2W] -> [B, cout, D, H, W]
    x = torch.randn(B, cin, 2 * D, 2 * H, 2 * W, generator=gen)
    w = torch.randn(cout, cin, 2, 2, 2, generator=gen) / (8 * cin) ** 0.5
    ref = torch.nn.functional.conv3d(x.double(), w.double(), stride=2)
    gi, go = geom(B * (2 * D + 2), 2 * H, 2 * W), geom(B * (D + 2), H, W)
    xa = vol_to_act(x, gi)
    ya = torch.zeros((cout // 8, go.cs, 8))
    for dz in range(2):
        wp = pack_down_bf16s_weight(w[:, :, dz].contiguous())
        E.check(l.dinv_conv_down2x2_bf16s_3d(ctypes.byref(gi), ctypes.byref(go), E.p(xa), ctypes.c_void_p(wp.data_ptr()), cin,
                                             cout, E.p(ya), D, dz, int(dz > 0), None))
    out = act_to_vol(ya, go, cout, B, D)
    assert float(out[:, 0].abs().max()) == 0 and float(out[:, D + 1].abs().max()) == 0        # zero slices stay zero
    got = out[:, 1:-1].permute(0, 2, 1, 3, 4)
    assert float((got.double() - ref).norm() / ref.norm()) < 2e-5
    # ---- up: [B, cin, D, H, W] -> [B, cout, 2D, 2H, 2W]
    x = torch.randn(B, cin, D, H, W, generator=gen)
    w = torch.randn(cin, cout, 2, 2, 2, generator=gen) / cin ** 0.5
    ref = torch.nn.functional.conv_transpose3d(x.double(), w.double(), stride=2)
    xa = vol_to_act(x, go)
    ya = torch.zeros((cout // 8, gi.cs, 8))
    for dz in range(2):
        wp = pack_up_bf16s_weight(w[:, :, dz].contiguous())
        E.check(l.dinv_conv_up2x2_bf16s_3d(ctypes.byref(go), ctypes.byref(gi), E.p(xa), None, ctypes.c_void_p(wp.data_ptr()), cin,
                                           cout, E.p(ya), D, dz, None))
    out = act_to_vol(ya, gi, cout, B, 2 * D)
    assert float(out[:, 0].abs().max()) == 0 and float(out[:, 2 * D + 1].abs().max()) == 0
    got = out[:, 1:-1].permute(0, 2, 1, 3, 4)
    assert float((got.double() - ref).norm() / ref.norm()) < 2e-5


@pytest.mark.parametrize("up", [False, True])
def test_wgrad_2x2x2_depth_tap(up):
    """dinv_conv_wgrad_3d: weight gradient of one depth tap of the 2x2x2 stride-2 (transposed) convolution vs autograd"""
    B, D, H, W, cs, cl = 2, 2, 4, 4, 32, 16
    gen = torch.Generator().manual_seed(int(up) + 5)
    small = torch.randn(B, cs, D, H, W, generator=gen)
    big = torch.randn(B, cl, 2 * D, 2 * H, 2 * W, generator=gen)
    w = torch.zeros(cs, cl, 2, 2, 2, dtype=torch.float64, requires_grad=True)
    if up:
        (torch.nn.functional.conv_transpose3d(small.double(), w, stride=2) * big.double()).sum().backward()
    else:
        (torch.nn.functional.conv3d(big.double(), w, stride=2) * small.double()).sum().backward()
    gs, gl = geom(B * (D + 2), H, W), geom(B * (2 * D + 2), 2 * H, 2 * W)
    sa, la = vol_to_act(small, gs), vol_to_act(big, gl)
    lib = E.lib()
    lib.dinv_conv_wgrad_workspace_bytes.restype = ctypes.c_size_t
    for dz in range(2):
        dw = torch.full((cs, cl, 2, 2), float("nan"))
        ws = torch.zeros(lib.dinv_conv_wgrad_workspace_bytes(ctypes.byref(gs), cs, cl, 4), dtype=torch.uint8)
        E.check(lib.dinv_conv_wgrad_3d(ctypes.byref(gs), ctypes.byref(gl), E.p(sa), cs, E.p(la), cl, E.p(dw), 0, E.p(ws),
                                       ctypes.c_size_t(ws.numel()), D, dz, None))
        assert float((dw.double() - w.grad[:, :, dz]).norm() / w.grad[:, :, dz].norm()) < 1e-5


@pytest.mark.parametrize("B,C,D,H,W,cout", [(2, 16, 3, 6, 16, 16), (1, 24, 2, 9, 10, 40), (2, 3, 4, 5, 7, 16), (1, 64, 2, 8, 8, 32)])
def test_wgrad_3x3x3_one_call_matches_autograd(B, C, D, H, W, cout):
    """dinv_conv_wgrad_3x3x3 (the three depth taps as the second grid dimension of one launch and of one reduction) against the
    autograd weight gradient of conv3d in fp64, and equal to three dinv_conv_wgrad calls on slice-shifted views bit for bit"""
    gen = torch.Generator().manual_seed(C + cout)
    x = torch.randn(B, C, D, H, W, generator=gen)
    gy = torch.randn(B, cout, D, H, W, generator=gen)
    w = torch.zeros(cout, C, 3, 3, 3, dtype=torch.float64, requires_grad=True)
    (torch.nn.functional.conv3d(x.double(), w, padding=1) * gy.double()).sum().backward()
    g = geom(B * (D + 2), H, W)
    guard = g.plane
    g.cs = (g.cs + 2 * guard + 3) // 4 * 4

    def to_vol(t):
        c = t.shape[1]
        cp = (c + 7) // 8 * 8
        a = torch.zeros(cp // 8, g.cs, 8)
        t2 = torch.nn.functional.pad(t.permute(0, 2, 1, 3, 4), (0, 0, 0, 0, 0, cp - c, 1, 1)).reshape(B * (D + 2), cp, H, W)
        fr = a[:, guard + g.sl: guard + g.sl + g.np].view(-1, B * (D + 2), g.hp, g.wp, 8)
        fr[:, :, 1:H + 1, 1:W + 1] = t2.reshape(B * (D + 2), -1, 8, H, W).permute(1, 0, 3, 4, 2)
        return a

    lib = E.lib()
    lib.dinv_conv_wgrad_workspace_bytes.restype = ctypes.c_size_t
    sa, la = to_vol(gy), to_vol(x)
    view = lambda a, dz=0: ctypes.c_void_p(a[:, guard + dz * g.plane:].data_ptr())
    dw = torch.full((cout, C, 3, 3, 3), float("nan"))
    nb = lib.dinv_conv_wgrad_workspace_bytes(ctypes.byref(g), cout, C, 9)
    ws = torch.zeros(3 * nb, dtype=torch.uint8)
    E.check(lib.dinv_conv_wgrad_3x3x3(ctypes.byref(g), view(sa), cout, view(la, -1), C, ctypes.c_int64(g.plane * 8), E.p(dw), 0, E.p(ws),
                                      ctypes.c_size_t(ws.numel()), None))
    assert not torch.isnan(dw).any()
    assert float((dw.double() - w.grad).norm() / w.grad.norm()) < 1e-5
    for dz in range(3):
        one = torch.full((cout, C, 3, 3), float("nan"))
        ws1 = torch.zeros(nb, dtype=torch.uint8)
        E.check(lib.dinv_conv_wgrad(ctypes.byref(g), ctypes.byref(g), view(sa), cout, view(la, dz - 1), C, 9, E.p(one), 0, E.p(ws1),
                                    ctypes.c_size_t(ws1.numel()), None))
        assert torch.equal(one, dw[:, :, dz])


@pytest.mark.parametrize("mode", ["plain", "relu_split_chain"])
def test_conv3x3x3_single_launch(mode):
    """dinv_conv3x3x3_split: the three depth taps inside the K loop of the 2-D-tile kernel, padding slices written as zeros;
    `relu_split_chain`: conv1 (ReLU, pre-split output) feeding conv2 (pre-split input + fp32 residual) like a 3-D ResBlock,
    against conv3d in fp64"""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from deepinv_amd.hip.drunet import pack_split3d_weight
    B, C, D, H, W, cout = 2, 16, 3, 6, 16, 64
    gen = torch.Generator().manual_seed(19)
    x = torch.randn(B, C, D, H, W, generator=gen)
    w1 = torch.randn(cout, C, 3, 3, 3, generator=gen) / (27 * C) ** 0.5
    w2 = torch.randn(cout, cout, 3, 3, 3, generator=gen) / (27 * cout) ** 0.5
    g = geom(B * (D + 2), H, W)
    guard = g.plane
    g.cs = (g.cs + 2 * guard + 3) // 4 * 4

    def to_vol(t):
        c = t.shape[1]
        a = torch.zeros(c // 8, g.cs, 8)
        t2 = torch.nn.functional.pad(t.permute(0, 2, 1, 3, 4), (0, 0, 0, 0, 0, 0, 1, 1)).reshape(B * (D + 2), c, H, W)
        fr = a[:, guard + g.sl: guard + g.sl + g.np].view(-1, B * (D + 2), g.hp, g.wp, 8)
        fr[:, :, 1:H + 1, 1:W + 1] = t2.reshape(B * (D + 2), -1, 8, H, W).permute(1, 0, 3, 4, 2)
        return a

    def from_vol(a, c):
        out = a[:, guard + g.sl: guard + g.sl + g.np].view(-1, B, D + 2, g.hp, g.wp, 8)
        assert float(out[:, :, 0].abs().max()) == 0 and float(out[:, :, D + 1].abs().max()) == 0     # padding slices
        return out[:, :, 1:-1, 1:H + 1, 1:W + 1].permute(1, 0, 5, 2, 3, 4).reshape(B, c, D, H, W)

    l = E.lib()
    xa = to_vol(x)
    view = lambda a: ctypes.c_void_p(a[:, guard:].data_ptr())
    p1, p2 = pack_split3d_weight(w1), pack_split3d_weight(w2)
    if mode == "plain":
        ya = torch.full((cout // 8, g.cs, 8), 7.0)      # stale values in the padding slices must be overwritten
        ya[:, :guard + g.sl] = 0
        ya[:, guard + g.sl + g.np:] = 0
        fr = ya[:, guard + g.sl: guard + g.sl + g.np].view(-1, B * (D + 2), g.hp, g.wp, 8)
        fr[:, :, :, 0] = 0
        fr[:, :, :, W + 1:] = 0
        E.check(l.dinv_conv3x3x3_split(ctypes.byref(g), view(xa), ctypes.c_void_p(p1.data_ptr()), C, cout, view(ya), None, 0, D, None))
        ref = torch.nn.functional.conv3d(x.double(), w1.double(), padding=1)
        assert float((from_vol(ya, cout).double() - ref).norm() / ref.norm()) < 2e-5
        return
    ta, ya, ra = torch.zeros(cout // 8, g.cs, 8), torch.zeros(cout // 8, g.cs, 8), to_vol(torch.randn(B, cout, D, H, W, generator=gen))
    E.check(l.dinv_conv3x3x3_split(ctypes.byref(g), view(xa), ctypes.c_void_p(p1.data_ptr()), C, cout, view(ta), None, 2 | 4, D, None))
    E.check(l.dinv_conv3x3x3_split(ctypes.byref(g), view(ta), ctypes.c_void_p(p2.data_ptr()), cout, cout, view(ya), view(ra), 1, D, None))
    t_ref = torch.nn.functional.conv3d(x.double(), w1.double(), padding=1).relu()
    ref = torch.nn.functional.conv3d(t_ref, w2.double(), padding=1) + from_vol(ra, cout).double()
    assert float((from_vol(ya, cout).double() - ref).norm() / ref.norm()) < 3e-5


def test_conv3x3x3_as_three_shifted_2d_launches():
    """the composition models/drunet3d.py uses for a 3x3x3 convolution: three launches of the bf16-split 3x3 kernel on
    views of the input shifted by -1 / 0 / +1 slices (one guard plane in front of the buffer), accumulated IN PLACE
    through the residual input (res1 = y), against conv3d"""
    B, C, D, H, W, cout = 1, 16, 3, 6, 8, 64
    gen = torch.Generator().manual_seed(9)
    x = torch.randn(B, C, D, H, W, generator=gen)
    w = torch.randn(cout, C, 3, 3, 3, generator=gen) / (27 * C) ** 0.5
    ref = torch.nn.functional.conv3d(x.double(), w.double(), padding=1)
    g = geom(B * (D + 2), H, W)
    guard = g.plane
    g.cs = (g.cs + 2 * guard + 3) // 4 * 4                   # room for the slice-shifted views
    xa = torch.zeros(C // 8, g.cs, 8)
    t2 = torch.nn.functional.pad(x.permute(0, 2, 1, 3, 4), (0, 0, 0, 0, 0, 0, 1, 1)).reshape(B * (D + 2), C, H, W)
    frames = xa[:, guard + g.sl: guard + g.sl + g.np].view(-1, B * (D + 2), g.hp, g.wp, 8)
    frames[:, :, 1:H + 1, 1:W + 1] = t2.view(B * (D + 2), -1, 8, H, W).permute(1, 0, 3, 4, 2)
    ya = torch.zeros(cout // 8, g.cs, 8)
    l = E.lib()
    for dz in range(3):
        xv = xa[:, guard + (dz - 1) * g.plane:]
        yv = ya[:, guard:]
        wp = pack(w[:, :, dz].contiguous())
        E.check(l.dinv_conv3x3_split(ctypes.byref(g), ctypes.c_void_p(xv.data_ptr()), ctypes.c_void_p(wp.data_ptr()), C, cout,
                                     ctypes.c_void_p(yv.data_ptr()), ctypes.c_void_p(yv.data_ptr()) if dz else None, 0, None))
    out = ya[:, guard + g.sl: guard + g.sl + g.np].view(-1, B, D + 2, g.hp, g.wp, 8)[:, :, 1:-1, 1:H + 1, 1:W + 1]
    got = out.permute(1, 0, 5, 2, 3, 4).reshape(B, cout, D, H, W)
    assert float((got.double() - ref).norm() / ref.norm()) < 2e-5




@pytest.mark.parametrize("cin,cout,relu,res", [(16, 16, True, False), (3, 16, False, False), (16, 2, False, True), (16, 32, True, False),
                                              (24, 64, False, True), (16, 16, False, "gate"), (8, 3, False, "gate")])
def test_conv3x3x3_fp32_single_launch(cin, cout, relu, res):
    """dinv_conv3x3x3 (csrc/drunet.hip: fp32 MFMA, depth taps inside the K loop; cout <= 16: conv3_thin_kernel on the
    16x16x4 tile) against conv3d in fp64: ReLU / residual / gate (ReLU backward: y = r > 0 ? conv : 0) epilogues, zeroed padding
    slices, stale output overwritten"""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from deepinv_amd.hip.drunet import pack_conv3x3x3_weight
    B, D, H, W = 2, 3, 6, 15
    gen = torch.Generator().manual_seed(cin + 7 * cout)
    x = torch.randn(B, cin, D, H, W, generator=gen)
    w = torch.randn(cout, cin, 3, 3, 3, generator=gen) / (27 * cin) ** 0.5
    r = torch.randn(B, cout, D, H, W, generator=gen)
    g = geom(B * (D + 2), H, W)
    guard = g.plane
    g.cs = (g.cs + 2 * guard + 3) // 4 * 4

    def to_vol(t):
        c = t.shape[1]
        cp = (c + 7) // 8 * 8
        t = torch.cat((t, torch.zeros(B, cp - c, D, H, W)), 1)
        a = torch.zeros(cp // 8, g.cs, 8)
        t2 = torch.nn.functional.pad(t.permute(0, 2, 1, 3, 4), (0, 0, 0, 0, 0, 0, 1, 1)).reshape(B * (D + 2), cp, H, W)
        fr = a[:, guard + g.sl: guard + g.sl + g.np].view(-1, B * (D + 2), g.hp, g.wp, 8)
        fr[:, :, 1:H + 1, 1:W + 1] = t2.reshape(B * (D + 2), -1, 8, H, W).permute(1, 0, 3, 4, 2)
        return a

    xa, ra = to_vol(x), to_vol(r)
    cb = (cout + 7) // 8
    ya = torch.full((cb, g.cs, 8), 7.0)                # stale values everywhere the kernel must write
    ya[:, :guard + g.sl] = 0
    ya[:, guard + g.sl + g.np:] = 0
    wpk, cip, cop = pack_conv3x3x3_weight(w)
    view = lambda a: ctypes.c_void_p(a[:, guard:].data_ptr())
    E.check(E.lib().dinv_conv3x3x3(ctypes.byref(g), view(xa), ctypes.c_void_p(wpk.data_ptr()), cip, cop, cout, int(wpk.shape[4]),
                                   view(ya), view(ra) if res else None, int(relu) | (2 if res == "gate" else 0), D, None))
    out = ya[:, guard + g.sl: guard + g.sl + g.np].view(-1, B, D + 2, g.hp, g.wp, 8)
    assert float(out[:, :, 0].abs().max()) == 0 and float(out[:, :, D + 1].abs().max()) == 0          # padding slices
    assert float(out[:, :, :, 0].abs().max()) == 0 and float(out[:, :, :, :, 0].abs().max()) == 0     # frames
    assert float(out[:, :, :, :, W + 1:].abs().max()) == 0
    got = out[:, :, 1:-1, 1:H + 1, 1:W + 1].permute(1, 0, 5, 2, 3, 4).reshape(B, cb * 8, D, H, W)[:, :cout]
    ref = torch.nn.functional.conv3d(x.double(), w.double(), padding=1)
    if relu:
        ref = ref.clamp_min(0)
    if res == "gate":
        ref = ref * (r > 0)
        assert bool((got[r <= 0] == 0).all())
    elif res:
        ref = ref + r.double()
    assert float((got.double() - ref).norm() / ref.norm()) < 2e-6


@pytest.mark.parametrize("cout", [16, 32, 64])
def test_conv3x3_fp32_direct(cout):
    """dinv_conv3x3 (fp32 MFMA direct kernel; cout_tile 16 = the thin-layer kernel) against conv2d in fp64"""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from deepinv_amd.hip.drunet import pack_conv3x3_weight
    B, C, H, W = 2, 16, 9, 21
    gen = torch.Generator().manual_seed(cout)
    x = torch.randn(B, C, H, W, generator=gen)
    w = torch.randn(cout, C, 3, 3, generator=gen) / (9 * C) ** 0.5
    g = geom(B, H, W)
    if cout == 16:
        wpk = w.reshape(1, 16, C // 8, 8, 9).permute(0, 2, 4, 1, 3).contiguous()
        mt = 16
    else:
        wpk, _, _ = pack_conv3x3_weight(w)
        mt = int(wpk.shape[3])
    ya = torch.full((cout // 8, g.cs, 8), 7.0)
    ya[:, :g.sl] = 0
    ya[:, g.sl + g.np:] = 0
    xa = to_act(x, g)
    E.check(E.lib().dinv_conv3x3(ctypes.byref(g), E.p(xa), None, E.p(wpk), C, cout, cout, mt, E.p(ya), None, None, 0, None))
    ref = torch.nn.functional.conv2d(x.double(), w.double(), padding=1)
    assert float((from_act(ya, g, cout).double() - ref).norm() / ref.norm()) < 2e-6


@pytest.mark.parametrize("B,H,W,cin,cout,skip", [(2, 9, 21, 16, 2, True), (1, 19, 70, 8, 1, False), (1, 8, 130, 16, 3, True),
                                                 (3, 17, 62, 24, 4, True), (1, 5, 63, 8, 2, False)])
def test_tail_conv_lane_shift(B, H, W, cin, cout, skip):
    """dinv_conv3x3_tail (tail3x3_shift_kernel: one load per pixel, the column taps arrive through wave shifts; 8 output rows
    per lane, strips of <= 62 columns per wave) against conv2d in fp64: one / two / three strips, ragged row groups, with and
    without the skip tensor added on the fly; the frame and the unused output channels stay untouched"""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from deepinv_amd.hip.drunet import pack_tail_weight
    gen = torch.Generator().manual_seed(W)
    x = torch.randn(B, cin, H, W, generator=gen)
    x2 = torch.randn(B, cin, H, W, generator=gen)
    w = torch.randn(cout, cin, 3, 3, generator=gen) / (9 * cin) ** 0.5
    g = geom(B, H, W)
    ya = torch.full((1, g.cs, 8), 7.0)
    xa, x2a, wpk = to_act(x, g), to_act(x2, g), pack_tail_weight(w)
    E.check(E.lib().dinv_conv3x3_tail(ctypes.byref(g), E.p(xa), E.p(x2a) if skip else None, E.p(wpk), cin, cout, E.p(ya), None))
    ref = torch.nn.functional.conv2d((x + x2 if skip else x).double(), w.double(), padding=1)
    yv = ya[:, g.sl:g.sl + g.np].view(1, B, g.hp, g.wp, 8)
    got = yv[0, :, 1:H + 1, 1:W + 1, :cout].permute(0, 3, 1, 2)
    assert float((got.double() - ref).norm() / ref.norm()) < 2e-6
    assert float(yv[0, :, 1:H + 1, 1:W + 1, 4:].min()) == 7.0 and float(yv[0, :, 0].min()) == 7.0      # untouched
    assert float(yv[0, :, :, 0].min()) == 7.0 and float(yv[0, :, :, W + 1:].min()) == 7.0
    if cout < 4:
        assert float(yv[0, :, 1:H + 1, 1:W + 1, cout:4].abs().max()) == 0.0     # the float4 store pads with zeros


@pytest.mark.parametrize("B,H,W,cin,cout,mode", [(1, 16, 32, 16, 64, "plain"), (2, 8, 8, 32, 64, "relu"), (1, 16, 16, 32, 128, "res"),
                                                 (3, 8, 12, 16, 64, "res"), (1, 20, 36, 16, 64, "relu"), (5, 8, 8, 16, 64, "plain"),
                                                 (20, 16, 16, 16, 128, "res"), (2, 32, 64, 48, 64, "relu"),
                                                 (3, 16, 16, 176, 128, "plain")])     # > 3 MB of U: cout tile outermost
@pytest.mark.parametrize("split", [False, True])
@pytest.mark.parametrize("bf3", [False, True])
def test_winograd4_conv_matches_fp64(B, H, W, cin, cout, mode, split, bf3):
    """csrc/drunet_wino4.hip (Winograd F(4x4,3x3) on the fp32 matrix cores: U fragments straight from memory, V through LDS,
    wave = 9 points of a cout half, two-round exchange in the epilogue) against an fp64 convolution: every rectangle shape
    (4x8 / 4x4 / 2x2 tiles), partial rectangles (W = 12, 36, H = 20), position groups that straddle images, several tiles per
    workgroup (the emulated device has 2 compute units per XCD), two cout tiles; with a workspace the tiles of the last
    incomplete round are cut along the input channels (partial outputs + ticket + ordered combine)"""
    gen = torch.Generator().manual_seed(H * W + cin)
    x = torch.randn(B, cin, H, W, generator=gen)
    w = torch.randn(cout, cin, 3, 3, generator=gen) / (3.0 * cin ** 0.5)
    r = torch.randn(B, cout, H, W, generator=gen)
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from deepinv_amd.hip.drunet import pack_winograd4_weight, pack_winograd4_bf16x3_weight
    g = geom(B, H, W)
    xa = to_act(x, g)
    ref = torch.nn.functional.conv2d(x.double(), w.double(), padding=1)
    if mode == "relu":
        ref = ref.relu()
    if mode == "res":
        ref = ref + r.double()
    ra = to_act(r, g)
    ya = torch.zeros((cout // 8, g.cs, 8))
    ya[:, g.sl:g.sl + g.np].view(-1, B, g.hp, g.wp, 8)[:, :, 1:H + 1, 1:W + 1] = float("nan")   # only interiors are written
    wp = pack_winograd4_bf16x3_weight(w) if bf3 else pack_winograd4_weight(w)    # (bf3: U split into three bf16 parts on the host)
    l = E.lib()
    l.dinv_conv3x3_winograd4_workspace_bytes.restype = ctypes.c_size_t
    ws = torch.zeros(l.dinv_conv3x3_winograd4_workspace_bytes(), dtype=torch.uint8) if split else None
    for _ in range(2 if split else 1):      # twice: the second launch finds the ticket words reset by the first
        if split:
            ya[:, g.sl:g.sl + g.np].view(-1, B, g.hp, g.wp, 8)[:, :, 1:H + 1, 1:W + 1] = float("nan")
        fn = l.dinv_conv3x3_winograd4_bf16x3 if bf3 else l.dinv_conv3x3_winograd4   # bf3: three-part bf16 split, six products
        E.check(fn(ctypes.byref(g), E.p(xa), ctypes.c_void_p(wp.data_ptr()), cin, cout, E.p(ya),
                                         E.p(ra) if mode == "res" else None, 1 if mode == "relu" else 0, E.p(ws),
                                         ctypes.c_size_t(0 if ws is None else ws.numel()), None))
    assert not torch.isnan(ya).any()
    f, nt = ctypes.c_int32(0), ctypes.c_int32(0)
    E.check(l.dinv_conv3x3_winograd4_last_split(ctypes.byref(f), ctypes.byref(nt)))
    if split:
        assert int(ws[:8 * 64 * 4].view(torch.int32).abs().max()) == 0      # the last arriver of every tile reset its ticket
        assert (f.value > 1) == (nt.value > 0)
    else:
        assert f.value == 1 and nt.value == 0
    out = from_act(ya, g, cout)
    err = float((out.double() - ref).norm() / ref.norm())
    assert err < 1e-5, err
    full = ya[:, g.sl:g.sl + g.np].view(-1, B, g.hp, g.wp, 8)
    assert float(full[:, :, 0].abs().max()) == 0 and float(full[:, :, H + 1].abs().max()) == 0
    assert float(full[:, :, :, 0].abs().max()) == 0 and float(full[:, :, :, W + 1:].abs().max()) == 0


def test_winograd4_rejects_bad_shapes():
    g = geom(1, 10, 16)
    x = torch.zeros(2, g.cs, 8)
    y = torch.zeros(8, g.cs, 8)
    w = torch.zeros(64 * 16 * 36)
    l = E.lib()
    assert l.dinv_conv3x3_winograd4(ctypes.byref(g), E.p(x), E.p(w), 16, 64, E.p(y), None, 0, None, ctypes.c_size_t(0), None) != 0      # height % 4
    g = geom(1, 8, 16)
    assert l.dinv_conv3x3_winograd4(ctypes.byref(g), E.p(x), E.p(w), 24, 64, E.p(y), None, 0, None, ctypes.c_size_t(0), None) != 0      # cin % 16


from winograd_ref import winograd4_magnitude  # noqa: E402


@pytest.mark.parametrize("case", ["wide", "he_scale"])
def test_winograd4_bf16x3_worst_case(case):
    """dinv_conv3x3_winograd4_bf16x3 (three-part bf16 split of BOTH operands, six products) element by element against the fp64
    convolution, next to the fp32-MFMA form of the same kernel, in units of winograd4_magnitude().  A bf16 part carries 8
    significand bits, so |xm| <= 2^-8 |x| and |xl| <= 2^-16 |x|: the three products the kernel drops (um vl, ul vm, ul vl) are
    below 2^-23 |u||v| in the worst case (both operands at half-ulp extremes; ~2^-26 rms), the split itself is exact to 2^-24:
    bound 2^-22 of the magnitude sum, 64 x 3 times tighter than the two-part split's 3 * 2^-16 (test_wsplit_worst_case), measured
    2^-23.2 for operands spanning 2^-20 .. 2^8 and 2^-12 .. 2^2 (the fp32 form: 2^-24.5); He-scaled weights on N(0,1) data stay
    below the fp32 form's 3.2e-6"""
    from deepinv_amd.hip.drunet import pack_winograd4_weight, pack_winograd4_bf16x3_weight
    gen = torch.Generator().manual_seed(21)
    B, H, W, cin, cout = 1, 16, 24, 32, 64

    def wide(shape, lo, hi):
        e = torch.randint(lo, hi + 1, shape, generator=gen).float()
        return (1 + torch.rand(shape, generator=gen)) * torch.exp2(e) * (torch.randint(0, 2, shape, generator=gen) * 2 - 1).float()

    if case == "wide":
        x, w = wide((B, cin, H, W), -20, 8), wide((cout, cin, 3, 3), -12, 2)
    else:
        x, w = torch.randn(B, cin, H, W, generator=gen), torch.randn(cout, cin, 3, 3, generator=gen) * (2.0 / (9 * cin)) ** 0.5
    ref = torch.nn.functional.conv2d(x.double(), w.double(), padding=1)
    mag = winograd4_magnitude(x, w)
    g = geom(B, H, W)
    xa = to_act(x, g)
    l = E.lib()
    worst = {}
    for name, fn, wp in (("fp32", l.dinv_conv3x3_winograd4, pack_winograd4_weight(w)),
                         ("bf16x3", l.dinv_conv3x3_winograd4_bf16x3, pack_winograd4_bf16x3_weight(w))):
        ya = torch.zeros((cout // 8, g.cs, 8))
        E.check(fn(ctypes.byref(g), E.p(xa), ctypes.c_void_p(wp.data_ptr()), cin, cout, E.p(ya), None, 0, None, ctypes.c_size_t(0), None))
        out = from_act(ya, g, cout).double()
        worst[name] = float(((out - ref).abs() / mag).max())
        assert worst[name] < 2.0 ** -22, (name, worst[name])
        if case == "he_scale":
            assert float((out - ref).norm() / ref.norm()) < 3.2e-6
