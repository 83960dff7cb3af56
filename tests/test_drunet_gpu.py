"""DRUNet MFMA path vs the CPU oracle (same seeded weights via state_dict)."""
import os

import pytest
import torch

from conftest import rel_err
from oracle import drunet_cpu as OD

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("cin,cout,B,H,W", [(2, 2, 2, 64, 64), (1, 1, 1, 96, 40), (3, 3, 2, 32, 72), (2, 2, 1, 320, 320)])
def test_drunet_matches_oracle(dev, cin, cout, B, H, W):
    import deepinv_amd as dinv

    sd = OD.init_state_dict(cin, cout, seed=1)
    model = dinv.models.DRUNet(cin, cout, pretrained=None).to(dev)
    model.load_state_dict(sd)
    model.eval()
    g = torch.Generator().manual_seed(0)
    x = torch.rand(B, cin, H, W, generator=g)
    ref = OD.drunet(sd, x, 0.05)
    with torch.no_grad():
        out = model(x.to(dev), 0.05)
    assert out.shape == ref.shape
    assert rel_err(out, ref) < 1e-4
    # per-sample sigma tensor
    s = torch.tensor([0.03 + 0.02 * i for i in range(B)])
    with torch.no_grad():
        out2 = model(x.to(dev), s.to(dev))
    assert rel_err(out2, OD.drunet(sd, x, s)) < 1e-4


def test_drunet_torch_training_path_matches_hip(dev):
    import deepinv_amd as dinv

    model = dinv.models.DRUNet(2, 2, pretrained=None).to(dev)
    x = torch.rand(1, 2, 64, 64, device=dev)
    with torch.no_grad():
        a = model(x, 0.1)
    xin = torch.cat((x, torch.full((1, 1, 64, 64), 0.1, device=dev)), 1)
    from torch_drunet import forward_unet_torch
    b = forward_unet_torch(model, xin)
    assert rel_err(a, b) < 1e-4


def test_drunet_unsafe_shapes(dev):
    import deepinv_amd as dinv

    model = dinv.models.DRUNet(1, 1, pretrained=None).to(dev).eval()
    with torch.no_grad():
        y = model(torch.rand(1, 1, 37, 41, device=dev), 0.1)    # test_pad branch
        z = model(torch.rand(1, 1, 70, 100, device=dev), 0.1)   # test_onesplit branch
    assert y.shape == (1, 1, 37, 41) and z.shape == (1, 1, 70, 100)
    assert torch.isfinite(y).all() and torch.isfinite(z).all()


@pytest.mark.parametrize("B,H,W,cin,cout", [(2, 64, 64, 64, 64), (3, 40, 40, 128, 128), (1, 80, 80, 64, 128),
                                              (2, 20, 20, 512, 512), (1, 37, 51, 64, 64), (5, 10, 10, 64, 64),
                                              (1, 320, 320, 64, 64),
                                              # smallest channel counts, one-pixel images, many tiny images
                                              (1, 24, 40, 32, 64), (2, 16, 16, 48, 64), (3, 1, 1, 64, 64), (70, 4, 6, 32, 128),
                                              (1, 2, 330, 64, 64)])
@pytest.mark.parametrize("mode", ["plain", "relu", "res"])
def test_winograd_conv_matches_fp32_conv(dev, B, H, W, cin, cout, mode):
    """Winograd F(2x2,3x3) ResBlock convolution against torch's fp32 conv2d of the same op (tolerance 1e-5:
    the transforms only add a few fp32 roundings) and against the direct MFMA kernel."""
    from deepinv_amd.hip import drunet as K

    g = torch.Generator().manual_seed(B * 1000 + H)
    x = torch.randn(B, cin, H, W, generator=g).to(dev)
    w = (torch.randn(cout, cin, 3, 3, generator=g) / (3.0 * cin ** 0.5)).to(dev)
    r = torch.randn(B, cout, H, W, generator=g).to(dev)
    ref = torch.nn.functional.conv2d(x.double(), w.double(), padding=1)
    if mode == "relu":
        ref = ref.relu()
    if mode == "res":
        ref = ref + r.double()

    geo = K.geom(B, H, W)

    def to_act(t):
        a = K.alloc(geo, t.shape[1], dev)
        av = a[:, geo.sl:geo.sl + geo.np].view(-1, B, geo.hp, geo.wp, 8)
        av[:, :, 1:H + 1, 1:W + 1] = t.view(B, -1, 8, H, W).permute(1, 0, 3, 4, 2)
        return a

    def from_act(a, c):
        av = a[:, geo.sl:geo.sl + geo.np].view(-1, B, geo.hp, geo.wp, 8)
        return av[:, :, 1:H + 1, 1:W + 1].permute(1, 0, 4, 2, 3).reshape(B, c, H, W)

    xa, ra = to_act(x), to_act(r)
    ya, yd = K.alloc(geo, cout, dev), K.alloc(geo, cout, dev)
    K.conv3x3_winograd(geo, xa, K.pack_winograd_weight(w), cin, cout, ya, res1=ra if mode == "res" else None,
                       relu=mode == "relu")
    wd, ci, co = K.pack_conv3x3_weight(w)
    K.conv3x3(geo, xa, wd, ci, co, yd, res1=ra if mode == "res" else None, relu=mode == "relu")
    out, direct = from_act(ya, cout), from_act(yd, cout)
    assert rel_err(out, ref) < 1e-5
    assert rel_err(out, direct) < 1e-5
    # the zero border of the padded frame must stay zero (the next layer reads it as padding)
    full = ya[:, geo.sl:geo.sl + geo.np].view(-1, B, geo.hp, geo.wp, 8)
    assert full[:, :, 0].abs().max() == 0 and full[:, :, H + 1:].abs().max() == 0
    assert full[:, :, :, 0].abs().max() == 0 and full[:, :, :, W + 1:].abs().max() == 0


def test_winograd_conv_rejects_unsupported_channel_counts(dev):
    """cin must be a multiple of 16 and >= 32 (channel blocks are pipelined in pairs, the last four peeled)"""
    from deepinv_amd.hip import drunet as K

    geo = K.geom(1, 8, 8)
    x, y = K.alloc(geo, 24, dev), K.alloc(geo, 64, dev)
    with pytest.raises(ValueError):
        K.pack_winograd_weight(torch.zeros(64, 24, 3, 3, device=dev))
    with pytest.raises(RuntimeError, match="winograd conv needs"):
        K.conv3x3_winograd(geo, x, torch.zeros(1, 3, 8, 64, 16, device=dev), 24, 64, y)


@pytest.mark.parametrize("B,H,W,cin,cout", [(2, 64, 64, 64, 64), (3, 40, 40, 128, 128), (1, 80, 80, 64, 128), (2, 20, 20, 512, 512),
                                              (1, 36, 52, 64, 64), (5, 12, 12, 64, 64), (1, 320, 320, 64, 64), (1, 24, 40, 32, 64),
                                              (2, 16, 16, 48, 64), (3, 4, 4, 16, 64), (70, 4, 8, 32, 128), (1, 4, 332, 64, 64),
                                              (32, 40, 40, 64, 128)])
@pytest.mark.parametrize("mode", ["plain", "relu", "res"])
def test_winograd4_conv_matches_fp64_conv(dev, B, H, W, cin, cout, mode):
    """Winograd F(4x4,3x3) ResBlock convolution on the fp32 matrix cores (csrc/drunet_wino4.hip) against an fp64 conv2d of the
    same op (1e-5: fp32 multiplies, the transforms add a few fp32 roundings, 2-3e-6 measured) and against the direct MFMA
    kernel: every rectangle shape, partial rectangles, position groups that straddle images, several tiles per workgroup"""
    from deepinv_amd.hip import drunet as K

    g = torch.Generator().manual_seed(B * 1000 + H)
    x = torch.randn(B, cin, H, W, generator=g).to(dev)
    w = (torch.randn(cout, cin, 3, 3, generator=g) / (3.0 * cin ** 0.5)).to(dev)
    r = torch.randn(B, cout, H, W, generator=g).to(dev)
    ref = torch.nn.functional.conv2d(x.double(), w.double(), padding=1)
    if mode == "relu":
        ref = ref.relu()
    if mode == "res":
        ref = ref + r.double()
    geo = K.geom(B, H, W)

    def to_act(t):
        a = K.alloc(geo, t.shape[1], dev)
        av = a[:, geo.sl:geo.sl + geo.np].view(-1, B, geo.hp, geo.wp, 8)
        av[:, :, 1:H + 1, 1:W + 1] = t.view(B, -1, 8, H, W).permute(1, 0, 3, 4, 2)
        return a

    def from_act(a, c):
        av = a[:, geo.sl:geo.sl + geo.np].view(-1, B, geo.hp, geo.wp, 8)
        return av[:, :, 1:H + 1, 1:W + 1].permute(1, 0, 4, 2, 3).reshape(B, c, H, W)

    xa, ra = to_act(x), to_act(r)
    ya, yd = K.alloc(geo, cout, dev), K.alloc(geo, cout, dev)
    K.conv3x3_winograd4(geo, xa, K.pack_winograd4_weight(w), cin, cout, ya, res1=ra if mode == "res" else None, relu=mode == "relu")
    wd, ci, co = K.pack_conv3x3_weight(w)
    K.conv3x3(geo, xa, wd, ci, co, yd, res1=ra if mode == "res" else None, relu=mode == "relu")
    out, direct = from_act(ya, cout), from_act(yd, cout)
    assert rel_err(out, ref) < 1e-5
    assert rel_err(out, direct) < 1e-5
    full = ya[:, geo.sl:geo.sl + geo.np].view(-1, B, geo.hp, geo.wp, 8)     # the zero frame is never written
    assert full[:, :, 0].abs().max() == 0 and full[:, :, H + 1:].abs().max() == 0
    assert full[:, :, :, 0].abs().max() == 0 and full[:, :, :, W + 1:].abs().max() == 0


@pytest.mark.parametrize("B,H,W,cin,cout", [(4, 320, 320, 64, 64), (4, 160, 160, 128, 128), (4, 80, 80, 256, 256), (4, 40, 40, 512, 512),
                                              (32, 40, 40, 512, 512), (1, 36, 52, 64, 64), (3, 4, 4, 16, 64), (2, 16, 16, 48, 64)])
@pytest.mark.parametrize("mode", ["relu", "res"])
def test_winograd4_bf16x3_is_fp32_equivalent(dev, B, H, W, cin, cout, mode):
    """dinv_conv3x3_winograd4_bf16x3 (csrc/drunet_wino4.hip, BF3: U pre-split on the host, V split in registers, three bf16 MFMAs per
    point and 8-channel block = six products) at the four DRUNet level shapes and the odd shapes of the fp32 form's test: per-layer
    error against the fp64 convolution at or below the fp32-MFMA form's (3.2e-6 is that form's worst level), with and without the
    tail split, and 20 back-to-back launches bit-identical"""
    from deepinv_amd.hip import drunet as K

    geo, xa, ra, wp, x, w, r = _w4_case(dev, B, H, W, cin, cout, seed=B + H + cin)
    w3 = K.pack_winograd4_bf16x3_weight(w)
    assert w3.dtype == torch.bfloat16 and w3.numel() * 2 == wp.numel() * 6
    res = ra if mode == "res" else None
    ref = torch.nn.functional.conv2d(x.double(), w.double(), padding=1)
    ref = ref.relu() if mode == "relu" else ref + r.double()

    def unpack(a):
        return a[:, geo.sl:geo.sl + geo.np].view(-1, B, geo.hp, geo.wp, 8)[:, :, 1:H + 1, 1:W + 1].permute(1, 0, 4, 2, 3).reshape(B, cout, H, W)

    y32, y3, y3s = (K.alloc(geo, cout, dev) for _ in range(3))
    K.conv3x3_winograd4(geo, xa, wp, cin, cout, y32, res1=res, relu=mode == "relu")
    K.conv3x3_winograd4_bf16x3(geo, xa, w3, cin, cout, y3, res1=res, relu=mode == "relu")
    ws = torch.zeros(K._l().dinv_conv3x3_winograd4_workspace_bytes(), device=dev, dtype=torch.uint8)
    K.conv3x3_winograd4_bf16x3(geo, xa, w3, cin, cout, y3s, res1=res, relu=mode == "relu", workspace=ws)
    e32, e3, e3s = (rel_err(unpack(a), ref) for a in (y32, y3, y3s))
    # (3.2e-6: the fp32 form's worst level without the ReLU; behind a ReLU the same absolute error meets half the norm - there
    # the fp32 form's own error on the same data is the bar)
    assert e3 < max(3.2e-6, e32) and e3s < max(3.2e-6, e32), (e3, e3s, e32)
    assert e3 < 1.15 * e32 + 1e-7, (e3, e32)
    assert int(ws[:8 * 64 * 4].view(torch.int32).abs().max()) == 0
    full = y3[:, geo.sl:geo.sl + geo.np].view(-1, B, geo.hp, geo.wp, 8)     # the zero frame is never written
    assert full[:, :, 0].abs().max() == 0 and full[:, :, H + 1:].abs().max() == 0
    assert full[:, :, :, 0].abs().max() == 0 and full[:, :, :, W + 1:].abs().max() == 0
    for _ in range(20):
        y = K.alloc(geo, cout, dev)
        K.conv3x3_winograd4_bf16x3(geo, xa, w3, cin, cout, y, res1=res, relu=mode == "relu", workspace=ws)
        assert torch.equal(y, y3s)


@pytest.mark.parametrize("case", ["wide", "he_scale"])
def test_winograd4_bf16x3_worst_case_bound(dev, case):
    """the hardware twin of tests/test_emu_drunet.py::test_winograd4_bf16x3_worst_case (derivation there): element by element
    |y - y_exact| <= 2^-22 winograd4_magnitude() for activations spanning 2^-20 .. 2^8 and weights 2^-12 .. 2^2 - the two-part
    split's bound is 3 * 2^-16 (test_wsplit_worst_case_bound)"""
    from deepinv_amd.hip import drunet as K
    from winograd_ref import winograd4_magnitude

    gen = torch.Generator().manual_seed(21)
    B, H, W, cin, cout = 2, 64, 96, 128, 128

    def wide(shape, lo, hi):
        e = torch.randint(lo, hi + 1, shape, generator=gen).float()
        return (1 + torch.rand(shape, generator=gen)) * torch.exp2(e) * (torch.randint(0, 2, shape, generator=gen) * 2 - 1).float()

    if case == "wide":
        x, w = wide((B, cin, H, W), -20, 8), wide((cout, cin, 3, 3), -12, 2)
    else:
        x, w = torch.randn(B, cin, H, W, generator=gen), torch.randn(cout, cin, 3, 3, generator=gen) * (2.0 / (9 * cin)) ** 0.5
    mag = winograd4_magnitude(x, w).to(dev)
    x, w = x.to(dev), w.to(dev)
    ref = torch.nn.functional.conv2d(x.double(), w.double(), padding=1)
    geo = K.geom(B, H, W)
    xa, ya = K.alloc(geo, cin, dev), K.alloc(geo, cout, dev)
    xa[:, geo.sl:geo.sl + geo.np].view(-1, B, geo.hp, geo.wp, 8)[:, :, 1:H + 1, 1:W + 1] = x.view(B, -1, 8, H, W).permute(1, 0, 3, 4, 2)
    K.conv3x3_winograd4_bf16x3(geo, xa, K.pack_winograd4_bf16x3_weight(w), cin, cout, ya)
    av = ya[:, geo.sl:geo.sl + geo.np].view(-1, B, geo.hp, geo.wp, 8)
    out = av[:, :, 1:H + 1, 1:W + 1].permute(1, 0, 4, 2, 3).reshape(B, cout, H, W).double()
    assert bool(((out - ref).abs() <= 2.0 ** -22 * mag).all()), float(((out - ref).abs() / mag).max())
    if case == "he_scale":
        assert rel_err(out, ref) < 3.2e-6


def _w4_case(dev, B, H, W, cin, cout, seed):
    from deepinv_amd.hip import drunet as K

    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, cin, H, W, generator=g).to(dev)
    w = (torch.randn(cout, cin, 3, 3, generator=g) / (3.0 * cin ** 0.5)).to(dev)
    r = torch.randn(B, cout, H, W, generator=g).to(dev)
    geo = K.geom(B, H, W)

    def to_act(t):
        a = K.alloc(geo, t.shape[1], dev)
        a[:, geo.sl:geo.sl + geo.np].view(-1, B, geo.hp, geo.wp, 8)[:, :, 1:H + 1, 1:W + 1] = t.view(B, -1, 8, H, W).permute(1, 0, 3, 4, 2)
        return a

    return geo, to_act(x), to_act(r), K.pack_winograd4_weight(w), x, w, r


@pytest.mark.parametrize("B,H,W,cin,cout,parts", [(4, 40, 40, 128, 64, 8), (8, 40, 40, 64, 64, 4), (8, 40, 40, 32, 64, 2),
                                                   (32, 40, 40, 512, 512, 8), (32, 80, 80, 256, 256, 4)])
@pytest.mark.parametrize("mode", ["relu", "res"])
def test_winograd4_tail_split_direct(dev, B, H, W, cin, cout, parts, mode):
    """The tail split of the F(4x4) kernel (csrc/drunet_wino4.hip, SPLIT = true: the tiles of the last incomplete round cut into
    2 / 4 / 8 parts along the input channels, partial outputs + ticket + ordered combine - the only cross-workgroup
    release / acquire protocol in the product) called DIRECTLY: shapes that force every split factor (checked through
    dinv_conv3x3_winograd4_last_split), against the un-split launch of the same kernel (the parts are output-transformed
    before they are summed: 1e-5, as against the fp64 convolution), 100 back-to-back launches re-using one workspace (bit-identical every time), and
    the ticket words back at zero afterwards."""
    from deepinv_amd.hip import drunet as K

    geo, xa, ra, wp, x, w, r = _w4_case(dev, B, H, W, cin, cout, seed=B * 7 + cin)
    res = ra if mode == "res" else None
    y0, y1 = K.alloc(geo, cout, dev), K.alloc(geo, cout, dev)
    K.conv3x3_winograd4(geo, xa, wp, cin, cout, y0, res1=res, relu=mode == "relu")                # no workspace: never split
    assert K.winograd4_last_split()[0] == 1
    ws = torch.zeros(K._l().dinv_conv3x3_winograd4_workspace_bytes(), device=dev, dtype=torch.uint8)
    K.conv3x3_winograd4(geo, xa, wp, cin, cout, y1, res1=res, relu=mode == "relu", workspace=ws)
    f, ntail = K.winograd4_last_split()
    cus = torch.cuda.get_device_properties(dev).multi_processor_count
    if cus == 256:
        assert f == parts and ntail > 0, (f, ntail)
    else:
        assert f >= 1
    ref = torch.nn.functional.conv2d(x.double(), w.double(), padding=1)
    ref = ref.relu() if mode == "relu" else ref + r.double()
    got = y1[:, geo.sl:geo.sl + geo.np].view(-1, B, geo.hp, geo.wp, 8)[:, :, 1:H + 1, 1:W + 1].permute(1, 0, 4, 2, 3).reshape(B, cout, H, W)
    assert rel_err(got, ref) < 1e-5
    assert rel_err(y1, y0) < 1e-5     # the parts are output-transformed before they are summed: the rounding of F(4x4), 2e-6 measured
    tickets = ws[:8 * 64 * 4].view(torch.int32)
    assert int(tickets.abs().max()) == 0              # every ticket word reset by the last arriver
    for _ in range(100):                              # back to back on one stream, one workspace
        y2 = K.alloc(geo, cout, dev)
        K.conv3x3_winograd4(geo, xa, wp, cin, cout, y2, res1=res, relu=mode == "relu", workspace=ws)
        assert torch.equal(y2, y1)
    assert int(tickets.abs().max()) == 0


def test_winograd4_tail_split_two_streams(dev):
    """two convolutions with tail splits in flight on two streams at once, each with the workspace of its own stream
    (hip/drunet.py: winograd4_workspace is keyed by device AND stream): results bit-identical to the serial ones"""
    from deepinv_amd.hip import drunet as K

    cases = [_w4_case(dev, 4, 40, 40, 128, 64, seed=1), _w4_case(dev, 8, 40, 40, 64, 128, seed=2)]
    dims = [(128, 64), (64, 128)]
    serial = []
    for (geo, xa, ra, wp, *_), (ci, co) in zip(cases, dims):
        y = K.alloc(geo, co, dev)
        K.conv3x3_winograd4(geo, xa, wp, ci, co, y, res1=ra, workspace=K.winograd4_workspace(dev))
        assert K.winograd4_last_split()[0] > 1
        serial.append(y)
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
    outs = [[], []]
    for rep in range(20):
        for i, ((geo, xa, ra, wp, *_), (ci, co)) in enumerate(zip(cases, dims)):
            with torch.cuda.stream(streams[i]):
                y = K.alloc(geo, co, dev)
                ws = K.winograd4_workspace(dev)
                K.conv3x3_winograd4(geo, xa, wp, ci, co, y, res1=ra, workspace=ws)
                outs[i].append(y)
    torch.cuda.synchronize()
    with torch.cuda.stream(streams[0]):
        w0 = K.winograd4_workspace(dev)
    with torch.cuda.stream(streams[1]):
        w1 = K.winograd4_workspace(dev)
    assert w0.data_ptr() != w1.data_ptr() != K.winograd4_workspace(dev).data_ptr()
    for i in range(2):
        for y in outs[i]:
            assert torch.equal(y, serial[i])


@pytest.mark.parametrize("B,H,W,cin,cout,skip", [(2, 9, 21, 16, 2, True), (1, 19, 70, 8, 1, False), (1, 8, 130, 16, 3, True),
                                                 (3, 17, 62, 24, 4, True), (4, 320, 320, 64, 2, True), (1, 5, 63, 8, 2, False)])
def test_tail_conv_lane_shift(dev, B, H, W, cin, cout, skip):
    """dinv_conv3x3_tail (csrc/drunet_tail.hip tail3x3_shift_kernel: one load per pixel and tensor, column taps through
    v_mov_b32_dpp wave_shr / wave_shl, 8 output rows per lane, strips of <= 62 columns per wave) against an fp64 conv2d of
    x (+ x2): one to six strips, ragged row groups, the bench shape; frame and spare channels of the output untouched"""
    from deepinv_amd.hip import drunet as K

    g = torch.Generator().manual_seed(W)
    x = torch.randn(B, cin, H, W, generator=g).to(dev)
    x2 = torch.randn(B, cin, H, W, generator=g).to(dev)
    w = (torch.randn(cout, cin, 3, 3, generator=g) / (9 * cin) ** 0.5).to(dev)
    geo = K.geom(B, H, W)

    def to_act(t):
        a = K.alloc(geo, t.shape[1], dev)
        av = a[:, geo.sl:geo.sl + geo.np].view(-1, B, geo.hp, geo.wp, 8)
        av[:, :, 1:H + 1, 1:W + 1] = t.view(B, -1, 8, H, W).permute(1, 0, 3, 4, 2)
        return a

    xa, x2a, wpk = to_act(x), to_act(x2), K.pack_tail_weight(w)
    ya = torch.full((1, geo.cs, 8), 7.0, device=dev)
    K.conv3x3_tail(geo, xa, wpk, cin, cout, ya, x2=x2a if skip else None)
    ref = torch.nn.functional.conv2d((x + x2 if skip else x).double(), w.double(), padding=1)
    yv = ya[:, geo.sl:geo.sl + geo.np].view(1, B, geo.hp, geo.wp, 8)
    assert rel_err(yv[0, :, 1:H + 1, 1:W + 1, :cout].permute(0, 3, 1, 2), ref) < 2e-6
    assert float(yv[0, :, 1:H + 1, 1:W + 1, 4:].min()) == 7.0 and float(yv[0, :, 0].min()) == 7.0
    assert float(yv[0, :, :, 0].min()) == 7.0 and float(yv[0, :, :, W + 1:].min()) == 7.0
    assert float(yv[0, :, H + 1:].min()) == 7.0


@pytest.mark.parametrize("B,side,cout", [(8, 256, 1), (8, 256, 2), (8, 256, 3), (8, 256, 4), (32, 320, 2), (2, 320, 3)])
def test_tail_conv_reproducible_beside_a_bf16_split_launch(dev, B, side, cout):
    """The tail convolution on one stream while bf16-split convolutions (csrc/drunet_wsplit.hip) run on another - what the batch
    lanes of models/drunet.py do - returns the bits it returns alone.  Built with SLP-packed fp32 ops it did not (60 of 60
    launches: single terms dropped in lanes 48..63 of a wave, scripts/r06/race_hunt8.py / race_hunt11.py - the op_sel[1] = 1 forms of
    DESIGN.md 3.6); csrc/Makefile builds the library without them.  Row groups of 8, 4 and 2 rows per wave (launch sizes 32, 8 and 2 slices)."""
    from deepinv_amd.hip import drunet as K

    gen = torch.Generator().manual_seed(cout)
    geo, c = K.geom(B, side, side), 64

    def act(fill=True):
        a = K.alloc(geo, c, dev)
        if fill:
            t = torch.randn(B, c, side, side, generator=gen).to(dev)
            a[:, geo.sl:geo.sl + geo.np].view(-1, B, geo.hp, geo.wp, 8)[:, :, 1:side + 1, 1:side + 1] = \
                t.view(B, -1, 8, side, side).permute(1, 0, 3, 4, 2)
        return a

    xa, x2a, xb, rb, yb = act(), act(), act(), act(), act(False)
    wt = K.pack_tail_weight((torch.randn(cout, c, 3, 3, generator=gen) / 24).to(dev))
    wws = K.pack_wsplit_weight((torch.randn(c, c, 3, 3, generator=gen) / 24).to(dev))
    yt = K.alloc(geo, cout, dev)
    sa, sb = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    torch.cuda.synchronize()
    with torch.cuda.stream(sa):
        K.conv3x3_tail(geo, xa, wt, c, cout, yt, x2=x2a)
    torch.cuda.synchronize()
    alone = yt.clone()
    for _ in range(20):
        yt.zero_()
        for s, n in ((sb, 2), (sa, 0), (sb, 2)):
            with torch.cuda.stream(s):
                for _ in range(n):
                    K.conv3x3_wsplit(geo, xb, wws, c, c, yb, res1=rb)
                if not n:
                    K.conv3x3_tail(geo, xa, wt, c, cout, yt, x2=x2a)
        torch.cuda.synchronize()
        assert torch.equal(yt, alone)


@pytest.mark.parametrize("B,H,W,cin,cout", [(2, 16, 24, 16, 64), (3, 40, 40, 64, 128), (1, 64, 64, 256, 512), (32, 80, 80, 128, 256)])
def test_bf16x3_down2x2_is_fp32_equivalent(dev, B, H, W, cin, cout):
    """dinv_conv_down2x2_bf16x3 (three-part bf16 operand split, six products: the stride-2 layers of conv_precision = "fp32")
    against an fp64 convolution: at the level of the fp32-MFMA kernel it replaces (asserted < 1e-6 and within 3x of that
    kernel's own error; the two-part kernel of the bf16split setting sits at 3e-6)"""
    from deepinv_amd.hip import drunet as K

    g = torch.Generator().manual_seed(cin + H)
    x = torch.randn(B, cin, H, W, generator=g).to(dev)
    w = (torch.randn(cout, cin, 2, 2, generator=g) / (2.0 * cin ** 0.5)).to(dev)
    ref = torch.nn.functional.conv2d(x.double(), w.double(), stride=2)
    gi, go = K.geom(B, H, W), K.geom(B, H // 2, W // 2)
    xa = K.alloc(gi, cin, dev)
    xa[:, gi.sl:gi.sl + gi.np].view(-1, B, gi.hp, gi.wp, 8)[:, :, 1:H + 1, 1:W + 1] = x.view(B, -1, 8, H, W).permute(1, 0, 3, 4, 2)

    def run(fn, pack):
        ya = K.alloc(go, cout, dev)
        fn(gi, go, xa, pack(w), cin, cout, ya)
        return ya[:, go.sl:go.sl + go.np].view(-1, B, go.hp, go.wp, 8)[:, :, 1:H // 2 + 1, 1:W // 2 + 1].permute(1, 0, 4, 2, 3).reshape(ref.shape)

    e3 = rel_err(run(K.down2x2_bf16x3, K.pack_down_bf16x3_weight), ref)
    e32 = rel_err(run(K.down2x2, K.pack_down_weight), ref)
    e2 = rel_err(run(K.down2x2_bf16s, K.pack_down_bf16s_weight), ref)
    assert e3 < 1e-6 and e3 < 3 * e32 + 1e-7 and e2 > 3 * e3, (e3, e32, e2)


def test_winograd4_conv_rejects_unsupported_shapes(dev):
    from deepinv_amd.hip import drunet as K

    with pytest.raises(ValueError):
        K.pack_winograd4_weight(torch.zeros(64, 24, 3, 3, device=dev))
    geo = K.geom(1, 10, 8)
    x, y = K.alloc(geo, 16, dev), K.alloc(geo, 64, dev)
    with pytest.raises(RuntimeError, match="multiples of 4"):
        K.conv3x3_winograd4(geo, x, torch.zeros(64 * 16 * 36, device=dev), 16, 64, y)


@pytest.mark.parametrize("tile", [4, 2])
def test_drunet_fp32_precision_matches_oracle(dev, tile, monkeypatch):
    """the ONE precision switch: conv_precision = "fp32" (fp32 multiplies on the fp32 matrix cores: Winograd F(4x4,3x3), or
    F(2x2,3x3) / direct MFMA kernels); batch 32 so that every level clears the F(4x4) kernel's tile threshold"""
    import deepinv_amd as dinv
    from deepinv_amd.hip import drunet as K

    monkeypatch.setattr(K, "FP32_WINOGRAD_TILE", tile)
    sd = OD.init_state_dict(2, 2, seed=1)
    model = dinv.models.DRUNet(2, 2, pretrained=None).to(dev)
    model.load_state_dict(sd)
    model.eval()
    model.conv_precision = "fp32"
    x = torch.rand(32, 2, 64, 96, generator=torch.Generator().manual_seed(0))
    K.profile_begin()
    with torch.no_grad():
        out = model(x.to(dev), 0.05)
    prof = K.profile_end()
    assert ("conv3x3_wino4_kernel" in prof) == (tile == 4) and (tile == 4 or "conv3x3_wino_kernel" in prof)
    assert rel_err(out[:2], OD.drunet(sd, x[:2], 0.05)) < 1e-4


@pytest.mark.parametrize("B,H,W,cin,cout,mode", [(2, 24, 40, 64, 64, "plain"), (1, 17, 33, 32, 128, "relu"),
                                                 (2, 16, 16, 128, 64, "res"), (3, 5, 7, 16, 64, "plain"),
                                                 (1, 160, 160, 128, 128, "relu"), (2, 20, 20, 512, 512, "res"),
                                                 (70, 4, 6, 32, 128, "plain"), (1, 2, 330, 64, 64, "res")])
@pytest.mark.parametrize("fmt", ["f32", "in_split", "out_split"])
def test_split2d_conv_matches_fp64_conv(dev, B, H, W, cin, cout, mode, fmt):
    """two-part bf16 split on 2-D pixel tiles (csrc/drunet_split2d.hip): three products, a few 1e-6 per layer against fp64;
    tile widths 32 / 16 / 8 incl. partial column tiles, row tiles that straddle images, many tiny images, a single long
    row; fp32 and pre-split activation buffers (a pre-split input is what the kernel would split an fp32 one into)"""
    from deepinv_amd.hip import drunet as K

    if fmt == "out_split" and mode == "res":
        pytest.skip("a pre-split output carries no residual")
    g = torch.Generator().manual_seed(B * 100 + H)
    x = torch.randn(B, cin, H, W, generator=g).to(dev)
    w = (torch.randn(cout, cin, 3, 3, generator=g) / (3.0 * cin ** 0.5)).to(dev)
    r = torch.randn(B, cout, H, W, generator=g).to(dev)
    geo = K.geom(B, H, W)

    def to_act(t):
        a = K.alloc(geo, t.shape[1], dev)
        av = a[:, geo.sl:geo.sl + geo.np].view(-1, B, geo.hp, geo.wp, 8)
        av[:, :, 1:H + 1, 1:W + 1] = t.view(B, -1, 8, H, W).permute(1, 0, 3, 4, 2)
        return a

    def presplit(a):      # [C/8, cs, 8] fp32 -> the same bytes holding (8 bf16 high parts | 8 bf16 low parts)
        hi = a.bfloat16()
        lo = (a - hi.float()).bfloat16()
        return torch.cat((hi, lo), dim=-1).view(torch.float32).contiguous()

    def unsplit(a):
        h = a.view(torch.bfloat16).view(a.shape[0], a.shape[1], 16)
        return (h[..., :8].float() + h[..., 8:].float()).contiguous()

    xa, ra, ya = to_act(x), to_act(r), K.alloc(geo, cout, dev)
    if fmt == "in_split":
        xa = presplit(xa)
        av = unsplit(xa)[:, geo.sl:geo.sl + geo.np].view(-1, B, geo.hp, geo.wp, 8)
        x = av[:, :, 1:H + 1, 1:W + 1].permute(1, 0, 4, 2, 3).reshape(B, cin, H, W)     # the operand the kernel sees
    ref = torch.nn.functional.conv2d(x.double(), w.double(), padding=1)
    if mode == "relu":
        ref = ref.relu()
    if mode == "res":
        ref = ref + r.double()
    K.conv3x3_split(geo, xa, K.pack_split2d_weight(w), cin, cout, ya, res1=ra if mode == "res" else None, relu=mode == "relu",
                    x_presplit=fmt == "in_split", y_presplit=fmt == "out_split")
    if fmt == "out_split":
        ya = unsplit(ya)
    av = ya[:, geo.sl:geo.sl + geo.np].view(-1, B, geo.hp, geo.wp, 8)
    out = av[:, :, 1:H + 1, 1:W + 1].permute(1, 0, 4, 2, 3).reshape(B, cout, H, W)
    assert rel_err(out, ref) < 2e-5
    assert av[:, :, 0].abs().max() == 0 and av[:, :, H + 1:].abs().max() == 0
    assert av[:, :, :, 0].abs().max() == 0 and av[:, :, :, W + 1:].abs().max() == 0


@pytest.mark.parametrize("B,H,W,cin,cout,mode", [(2, 24, 40, 64, 64, "plain"), (1, 17, 34, 32, 128, "relu"),
                                                 (2, 16, 16, 128, 64, "res"), (3, 5, 6, 16, 64, "plain"),
                                                 (1, 160, 160, 128, 128, "relu"), (2, 20, 20, 512, 512, "res"),
                                                 (70, 4, 6, 32, 128, "plain"), (1, 2, 330, 64, 64, "res"),
                                                 (2, 320, 320, 64, 64, "res"), (1, 80, 80, 48, 64, "relu")])
def test_wsplit_conv_matches_fp64_conv(dev, B, H, W, cin, cout, mode):
    """Winograd F(2,3) along rows on the bf16 matrix cores with the two-part operand split (csrc/drunet_wsplit.hip): three
    products per Winograd multiply, a few 1e-6 per layer against fp64; tile widths 32 / 16 / 8 incl. partial column tiles, row
    tiles that straddle images, many tiny images, a single long row, an odd number of 16-channel steps"""
    from deepinv_amd.hip import drunet as K

    g = torch.Generator().manual_seed(B * 100 + H)
    x = torch.randn(B, cin, H, W, generator=g).to(dev)
    w = (torch.randn(cout, cin, 3, 3, generator=g) / (3.0 * cin ** 0.5)).to(dev)
    r = torch.randn(B, cout, H, W, generator=g).to(dev)
    geo = K.geom(B, H, W)

    def to_act(t):
        a = K.alloc(geo, t.shape[1], dev)
        av = a[:, geo.sl:geo.sl + geo.np].view(-1, B, geo.hp, geo.wp, 8)
        av[:, :, 1:H + 1, 1:W + 1] = t.view(B, -1, 8, H, W).permute(1, 0, 3, 4, 2)
        return a

    xa, ra, ya = to_act(x), to_act(r), K.alloc(geo, cout, dev)
    ya[:, geo.sl:geo.sl + geo.np].view(-1, B, geo.hp, geo.wp, 8)[:, :, 1:H + 1, 1:W + 1] = float("nan")   # every interior pixel is written
    ref = torch.nn.functional.conv2d(x.double(), w.double(), padding=1)
    if mode == "relu":
        ref = ref.relu()
    if mode == "res":
        ref = ref + r.double()
    K.conv3x3_wsplit(geo, xa, K.pack_wsplit_weight(w), cin, cout, ya, res1=ra if mode == "res" else None, relu=mode == "relu")
    av = ya[:, geo.sl:geo.sl + geo.np].view(-1, B, geo.hp, geo.wp, 8)
    out = av[:, :, 1:H + 1, 1:W + 1].permute(1, 0, 4, 2, 3).reshape(B, cout, H, W)
    assert rel_err(out, ref) < 2e-5
    assert av[:, :, 0].abs().max() == 0 and av[:, :, H + 1:].abs().max() == 0
    assert av[:, :, :, 0].abs().max() == 0 and av[:, :, :, W + 1:].abs().max() == 0


def test_wsplit_conv_rejects_odd_width(dev):
    from deepinv_amd.hip import drunet as K

    geo = K.geom(1, 8, 9)
    xa, ya = K.alloc(geo, 16, dev), K.alloc(geo, 64, dev)
    with pytest.raises(RuntimeError, match="even image width"):
        K.conv3x3_wsplit(geo, xa, K.pack_wsplit_weight(torch.zeros(64, 16, 3, 3, device=dev)), 16, 64, ya)


@pytest.mark.parametrize("case", ["wide", "he_scale"])
def test_split_worst_case_bound(dev, case):
    """the operand split's worst-case bound element by element on the hardware kernel (the emulated twin of this test,
    tests/test_emu_drunet.py::test_split_worst_case, explains it): |y - y_exact| <= (3 * 2^-16 + 2^-20) (|w| conv |x|) for
    activations spanning 2^-20 .. 2^8 and weights 2^-12 .. 2^2; He-scaled weights and N(0,1) data stay at a few 1e-6"""
    from deepinv_amd.hip import drunet as K

    gen = torch.Generator().manual_seed(11)
    B, H, W, cin, cout = 2, 64, 96, 128, 128

    def wide(shape, lo, hi):
        e = torch.randint(lo, hi + 1, shape, generator=gen).float()
        return (1 + torch.rand(shape, generator=gen)) * torch.exp2(e) * (torch.randint(0, 2, shape, generator=gen) * 2 - 1).float()

    if case == "wide":
        x, w = wide((B, cin, H, W), -20, 8), wide((cout, cin, 3, 3), -12, 2)
    else:
        x, w = torch.randn(B, cin, H, W, generator=gen), torch.randn(cout, cin, 3, 3, generator=gen) * (2.0 / (9 * cin)) ** 0.5
    x, w = x.to(dev), w.to(dev)
    ref = torch.nn.functional.conv2d(x.double(), w.double(), padding=1)
    mag = torch.nn.functional.conv2d(x.double().abs(), w.double().abs(), padding=1)
    geo = K.geom(B, H, W)
    xa, ya = K.alloc(geo, cin, dev), K.alloc(geo, cout, dev)
    xa[:, geo.sl:geo.sl + geo.np].view(-1, B, geo.hp, geo.wp, 8)[:, :, 1:H + 1, 1:W + 1] = x.view(B, -1, 8, H, W).permute(1, 0, 3, 4, 2)
    K.conv3x3_split(geo, xa, K.pack_split2d_weight(w), cin, cout, ya)
    av = ya[:, geo.sl:geo.sl + geo.np].view(-1, B, geo.hp, geo.wp, 8)
    out = av[:, :, 1:H + 1, 1:W + 1].permute(1, 0, 4, 2, 3).reshape(B, cout, H, W).double()
    assert bool(((out - ref).abs() <= (3 * 2.0 ** -16 + 2.0 ** -20) * mag).all()), float(((out - ref).abs() / mag).max())
    if case == "he_scale":
        assert rel_err(out, ref) < 5e-6


@pytest.mark.parametrize("case", ["wide", "he_scale"])
def test_wsplit_worst_case_bound(dev, case):
    """worst-case bound of the Winograd operand-split kernel on the hardware (emulated twin with the derivation:
    tests/test_emu_drunet.py::test_wsplit_worst_case): |y - y_exact| <= (3 * 2^-16 + 2^-20) sum_k (|U_k| conv |V_k|) over the
    three Winograd points of an output pixel, for activations spanning 2^-20 .. 2^8 and weights 2^-12 .. 2^2; He-scaled
    weights and N(0,1) data stay at a few 1e-6"""
    from deepinv_amd.hip import drunet as K

    gen = torch.Generator().manual_seed(12)
    B, H, W, cin, cout = 2, 64, 96, 128, 128

    def wide(shape, lo, hi):
        e = torch.randint(lo, hi + 1, shape, generator=gen).float()
        return (1 + torch.rand(shape, generator=gen)) * torch.exp2(e) * (torch.randint(0, 2, shape, generator=gen) * 2 - 1).float()

    if case == "wide":
        x, w = wide((B, cin, H, W), -20, 8), wide((cout, cin, 3, 3), -12, 2)
    else:
        x, w = torch.randn(B, cin, H, W, generator=gen), torch.randn(cout, cin, 3, 3, generator=gen) * (2.0 / (9 * cin)) ** 0.5
    x, w = x.to(dev), w.to(dev)
    ref = torch.nn.functional.conv2d(x.double(), w.double(), padding=1)
    g64 = w.double()
    U = torch.stack((g64[..., 0], (g64[..., 0] + g64[..., 1] + g64[..., 2]) / 2, (g64[..., 0] - g64[..., 1] + g64[..., 2]) / 2,
                     g64[..., 2])).abs()
    xp = torch.nn.functional.pad(x.double(), (1, 1, 1, 1))
    d0, d1, d2, d3 = xp[..., 0:W:2], xp[..., 1:W + 1:2], xp[..., 2:W + 2:2], xp[..., 3:W + 3:2]
    V = [(d0 - d2).abs(), (d1 + d2).abs(), (d2 - d1).abs(), (d1 - d3).abs()]
    M = [torch.nn.functional.conv2d(V[k], U[k].unsqueeze(-1)) for k in range(4)]
    mag = torch.stack((M[0] + M[1] + M[2], M[1] + M[2] + M[3]), -1).reshape(B, cout, H, W)
    geo = K.geom(B, H, W)
    xa, ya = K.alloc(geo, cin, dev), K.alloc(geo, cout, dev)
    xa[:, geo.sl:geo.sl + geo.np].view(-1, B, geo.hp, geo.wp, 8)[:, :, 1:H + 1, 1:W + 1] = x.view(B, -1, 8, H, W).permute(1, 0, 3, 4, 2)
    K.conv3x3_wsplit(geo, xa, K.pack_wsplit_weight(w), cin, cout, ya)
    av = ya[:, geo.sl:geo.sl + geo.np].view(-1, B, geo.hp, geo.wp, 8)
    out = av[:, :, 1:H + 1, 1:W + 1].permute(1, 0, 4, 2, 3).reshape(B, cout, H, W).double()
    assert bool(((out - ref).abs() <= (3 * 2.0 ** -16 + 2.0 ** -20) * mag).all()), float(((out - ref).abs() / mag).max())
    if case == "he_scale":
        assert rel_err(out, ref) < 8e-6


def test_drunet_default_precision_matches_oracle(dev):
    import deepinv_amd as dinv

    sd = OD.init_state_dict(2, 2, seed=1)
    model = dinv.models.DRUNet(2, 2, pretrained=None).to(dev)
    model.load_state_dict(sd)
    model.eval()
    assert model.conv_precision == "fp32"           # the reference's arithmetic type is the default; "bf16split" is opt-in
    x = torch.rand(2, 2, 64, 96, generator=torch.Generator().manual_seed(0))
    with torch.no_grad():
        out = model(x.to(dev), 0.05)
        assert rel_err(out, OD.drunet(sd, x, 0.05)) < 1e-5
        model.conv_precision = "bf16split"
        assert rel_err(model(x.to(dev), 0.05), OD.drunet(sd, x, 0.05)) < 1e-4
    model.conv_precision = "fp64"
    with pytest.raises(ValueError, match="conv_precision"):
        with torch.no_grad():
            model(x.to(dev), 0.05)


def test_tile_parallel_drunet_single_rank(dev):
    """DistributedProcessing (overlap tiling) around the HIP DRUNet on one rank: the four 96x96 windows of a 128x128 image
    ride ONE denoiser call as a batch and the assembled result equals denoising every reflect-padded window on its own
    and pasting its inner 64x64 patch (what the halo cannot see of DRUNet's receptive field is NOT tested here: with
    random-init weights a 16-pixel halo leaves ~10 % on the seams)"""
    import deepinv_amd as dinv
    from deepinv_amd.distributed import BatchParallelContext, DistributedProcessing

    torch.manual_seed(0)
    model = dinv.models.DRUNet(2, 2, pretrained=None).to(dev).eval()
    x = torch.rand(1, 2, 128, 128, generator=torch.Generator().manual_seed(1)).to(dev)
    calls = []

    def denoiser(z, sigma):
        calls.append(tuple(z.shape))
        return model(z, sigma)

    ctx = BatchParallelContext(device=dev)
    proc = DistributedProcessing(ctx, denoiser, strategy_kwargs={"patch_size": 64, "overlap": 16})
    with torch.no_grad():
        y = proc(x, 0.05)
        xp = torch.nn.functional.pad(x, (16, 16, 16, 16), mode="reflect")
        ref = torch.empty_like(x)
        for r in (0, 64):
            for c in (0, 64):
                ref[:, :, r:r + 64, c:c + 64] = model(xp[:, :, r:r + 96, c:c + 96].contiguous(), 0.05)[:, :, 16:80, 16:80]
    assert calls == [(4, 2, 96, 96)]
    assert y.shape == x.shape and torch.isfinite(y).all()
    assert rel_err(y, ref) < 1e-5


def _backend(model, mode):
    """mode "hip": the product; "torch": the PyTorch-ROCm graph of the same module (tests/torch_drunet.py)"""
    import contextlib

    from torch_drunet import torch_backend
    return torch_backend(model) if mode == "torch" else contextlib.nullcontext(model)


def _grad_run(model, x0, sig0, v, mode, monkeypatch=None):
    model.zero_grad()
    x = x0.clone().requires_grad_(True)
    sig = sig0.clone().requires_grad_(True)
    with _backend(model, mode):
        y = model(x, sig)
    (y * v).sum().backward()
    return y.detach(), x.grad, sig.grad, {n: p.grad.clone() for n, p in model.named_parameters()}


def _grad_problem(dev, gain):
    import deepinv_amd as dinv

    torch.manual_seed(0)
    model = dinv.models.DRUNet(2, 2, pretrained=None).to(dev)
    if gain != 0.2:
        with torch.no_grad():
            for n, p in model.named_parameters():
                if ".res." in n:
                    torch.nn.init.orthogonal_(p, gain=gain)
    B, H, W = 2, 48, 64
    g = torch.Generator().manual_seed(3)
    x0 = torch.rand(B, 2, H, W, generator=g).to(dev)
    sig0 = (0.05 + 0.1 * torch.rand(B, 1, H, W, generator=g)).to(dev)
    v = torch.randn(B, 2, H, W, generator=g).to(dev)
    return model, x0, sig0, v


def test_drunet_hip_backward_matches_autograd(dev, monkeypatch):
    """models/drunet_train.py (forward + backward on the HIP kernels: data gradients through the forward kernels with
    re-packed weights, weight gradients through dinv_conv_wgrad) against autograd through the PyTorch graph of the same
    module: output, gradient of the image, of the noise-level map (the trainable g_param of unfolded PnP) and of every
    convolution weight."""
    model, x0, sig0, v = _grad_problem(dev, 0.2)
    y_t, gx_t, gs_t, gw_t = _grad_run(model, x0, sig0, v, "torch", monkeypatch)
    y_h, gx_h, gs_h, gw_h = _grad_run(model, x0, sig0, v, "hip", monkeypatch)
    assert rel_err(y_h, y_t) < 1e-4
    assert rel_err(gx_h, gx_t) < 1e-4 and rel_err(gs_h, gs_t) < 1e-4
    worst = max((rel_err(gw_h[n], gw_t[n]), n) for n in gw_t)
    assert worst[0] < 1e-4, worst            # measured ~1e-5 (fp32 forward: same ReLU masks as the fp32 reference)
    # the inference kernels in the forward pass (bf16 split, a few 1e-6): data gradients stay fp32-class, a few ReLU
    # masks at |z| ~ 1e-6 flip and move single rows of single weight gradients (see models/drunet_train.py)
    model.train_forward_precision = "bf16split"
    y_b, gx_b, gs_b, gw_b = _grad_run(model, x0, sig0, v, "hip", monkeypatch)
    assert rel_err(y_b, y_t) < 1e-4 and rel_err(gx_b, gx_t) < 1e-4 and rel_err(gs_b, gs_t) < 1e-4
    errs = sorted(rel_err(gw_b[n], gw_t[n]) for n in gw_t)
    assert errs[len(errs) // 2] < 2e-3 and errs[-1] < 5e-2


def test_drunet_hip_backward_unit_gain_linearised(dev, monkeypatch):
    """O(1)-gain ResBlock weights, where the 56 ResBlock convolutions carry the gradient.  With unit gains a single ReLU
    mask that differs between two fp32 implementations (|z| ~ 1e-7) moves every upstream gradient by ~1e-4, which
    says nothing about the kernels; so the activation is taken out on BOTH sides (identity instead of ReLU) and the
    whole chain - data-gradient convolutions, 2x2 down / up gradients, all weight gradients - must agree to 1e-4."""
    from deepinv_amd.hip import drunet as K
    from deepinv_amd.models import drunet_train as T

    model, x0, sig0, v = _grad_problem(dev, 1.0)
    for m in model.modules():
        if hasattr(m, "res"):
            m.res[1] = torch.nn.Identity()
    conv3 = T._conv3
    monkeypatch.setattr(T, "_conv3", lambda g, w, x, relu=False, res1=None, fp32=False, flip=False, gate=None: conv3(g, w, x, False, res1, fp32, flip))
    monkeypatch.setattr(K, "relu_backward", lambda act, grad: grad)
    y_t, gx_t, gs_t, gw_t = _grad_run(model, x0, sig0, v, "torch", monkeypatch)
    y_h, gx_h, gs_h, gw_h = _grad_run(model, x0, sig0, v, "hip", monkeypatch)
    assert rel_err(y_h, y_t) < 1e-4
    assert rel_err(gx_h, gx_t) < 1e-4 and rel_err(gs_h, gs_t) < 1e-4
    worst = max((rel_err(gw_h[n], gw_t[n]), n) for n in gw_t)
    assert worst[0] < 1e-4, worst


def test_drunet_hip_backward_frozen_weights_and_unsafe_shape(dev):
    """frozen denoiser inside an unrolled loop: only the data gradient is needed (no weight-gradient launches), and the
    padded path for sizes that are not multiples of 8 (test_pad, drunet.py:266-287) differentiates through the pad"""
    import deepinv_amd as dinv

    torch.manual_seed(1)
    model = dinv.models.DRUNet(1, 1, pretrained=None).to(dev)
    for p in model.parameters():
        p.requires_grad_(False)
    x = torch.rand(1, 1, 45, 50, device=dev, requires_grad=True)
    y = model(x, 0.07)
    y.square().sum().backward()
    g_hip = x.grad.clone()
    x2 = x.detach().clone().requires_grad_(True)
    with _backend(model, "torch"):
        model(x2, 0.07).square().sum().backward()
    assert rel_err(g_hip, x2.grad) < 1e-4


def test_drunet3d_hip_matches_torch_graph(dev, monkeypatch):
    """DRUNet(dim=3) (BASELINE config 4's denoiser: nc = 16..128, nb = 1) on the 2-D kernels (models/drunet3d.py:
    volumes as stacks of slices, 3x3x3 = three accumulated 3x3 launches, 2x2x2 layers with in-kernel slice pairing)
    against the PyTorch graph of the same module (Conv3d / ConvTranspose3d): inference output, then output and all
    gradients (volume, noise map, every weight) with autograd"""
    import deepinv_amd as dinv

    torch.manual_seed(0)
    model = dinv.models.DRUNet(2, 2, nc=(16, 32, 64, 128), nb=1, pretrained=None, dim=3).to(dev)
    B, D, H, W = 1, 16, 32, 48
    g = torch.Generator().manual_seed(4)
    x0 = torch.rand(B, 2, D, H, W, generator=g).to(dev)
    sig0 = (0.05 + 0.1 * torch.rand(B, 1, D, H, W, generator=g)).to(dev)
    v = torch.randn(B, 2, D, H, W, generator=g).to(dev)
    with torch.no_grad():
        with _backend(model, "torch"):
            y_ref = model(x0, sig0)
        y_inf = model(x0, sig0)            # inference at the model's conv_precision (default: fp32 3x3x3 convolutions)
        model.conv_precision = "bf16split"
        y_split = model(x0, sig0)          # bf16-split kernels
        model.conv_precision = "fp32"
    assert rel_err(y_inf, y_ref) < 1e-5 and rel_err(y_split, y_ref) < 1e-4

    def run(mode):
        model.zero_grad()
        x = x0.clone().requires_grad_(True)
        sig = sig0.clone().requires_grad_(True)
        with _backend(model, mode):
            y = model(x, sig)
        (y * v).sum().backward()
        return y.detach(), x.grad, sig.grad, {n: p.grad.clone() for n, p in model.named_parameters()}

    y_t, gx_t, gs_t, gw_t = run("torch")
    y_h, gx_h, gs_h, gw_h = run("hip")
    assert rel_err(y_h, y_t) < 1e-4
    assert rel_err(gx_h, gx_t) < 1e-4 and rel_err(gs_h, gs_t) < 1e-4
    assert all(gw_h[n].shape == gw_t[n].shape for n in gw_t)
    worst = max((rel_err(gw_h[n], gw_t[n]), n) for n in gw_t)
    assert worst[0] < 2e-4, worst
    # recycled activation buffers (models/drunet3d.py: Vol free list): poison every interior voxel of every released
    # buffer - a kernel that did not overwrite all of them would leak NaNs - and require bit-identical results
    from deepinv_amd.models import drunet3d as M3

    assert M3._POOL, "the free list is empty after a training step"
    for (_, ch, b, d, h, w), bufs in M3._POOL.items():
        lv = M3.Level(b, d, h, w)
        for t in bufs:
            vol = t[:, lv.guard + lv.g.sl: lv.guard + lv.g.sl + lv.g.np].view(t.shape[0], b, d + 2, h + 2, -1, 8)
            vol[:(ch + 7) // 8, :, 1:-1, 1:h + 1, 1:w + 1] = float("nan")     # (blocks of padded channels stay zero)
    monkeypatch.setattr(M3, "CHECK_RECYCLED", True)    # every reuse asserts: nothing but interiors is non-zero
    y_h2, gx_h2, gs_h2, gw_h2 = run("hip")
    assert torch.equal(y_h2, y_h) and torch.equal(gx_h2, gx_h) and torch.equal(gs_h2, gs_h)
    assert all(torch.equal(gw_h2[n], gw_h[n]) for n in gw_h)
    with torch.no_grad():
        assert torch.equal(model(x0, sig0), y_inf)
        model(x0[:, :, :, :16].contiguous(), sig0[:, :, :, :16].contiguous())      # another problem shape (height 16 instead of 32) drops the old buffers
    assert M3._POOL and all(k[4] <= 16 for k in M3._POOL)
    M3.release_buffers()
    assert not M3._POOL


def test_drunet_hip_backward_padded_channels(dev, monkeypatch):
    """a 2-D DRUNet whose channel counts are not multiples of 64 (24, 48, 96, 160): zero-padded inside the training node"""
    import deepinv_amd as dinv

    torch.manual_seed(3)
    model = dinv.models.DRUNet(1, 1, nc=(24, 48, 96, 160), nb=2, pretrained=None).to(dev)
    g = torch.Generator().manual_seed(8)
    x0 = torch.rand(2, 1, 32, 40, generator=g).to(dev)
    sig0 = (0.05 + 0.1 * torch.rand(2, 1, 32, 40, generator=g)).to(dev)
    v = torch.randn(2, 1, 32, 40, generator=g).to(dev)
    y_t, gx_t, gs_t, gw_t = _grad_run(model, x0, sig0, v, "torch", monkeypatch)
    y_h, gx_h, gs_h, gw_h = _grad_run(model, x0, sig0, v, "hip", monkeypatch)
    assert rel_err(y_h, y_t) < 1e-4 and rel_err(gx_h, gx_t) < 1e-4 and rel_err(gs_h, gs_t) < 1e-4
    assert all(gw_h[n].shape == gw_t[n].shape for n in gw_t)
    worst = max((rel_err(gw_h[n], gw_t[n]), n) for n in gw_t)
    assert worst[0] < 2e-4, worst
