"""Host-side DRUNet engine logic (no GPU): weight packing layouts and the Winograd F(2x2,3x3) algebra the
kernel is built on (drunet_wino.hip)."""
import pytest
import torch


def test_winograd_weight_pack_and_algebra():
    """pack_winograd_weight layout [Cout/64][Cin/8][ci 8][co 64][16] holds U = G g G^T, and the F(2x2,3x3)
    identity the kernel relies on, Y = A^T [ sum_ci U .* (B^T d B) ] A, reproduces the 3x3 correlation exactly
    (fp64), including the row form used by the waves: t = d[ra] + sigma d[rb], then the column transform."""
    from deepinv_amd.hip.drunet import pack_winograd_weight

    g = torch.Generator().manual_seed(7)
    cout, cin = 128, 32
    w = torch.randn(cout, cin, 3, 3, generator=g, dtype=torch.float64)
    pk = pack_winograd_weight(w.float())
    assert pk.shape == (cout // 64, cin // 8, 8, 64, 16)
    G = torch.tensor([[1, 0, 0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0, 0, 1]], dtype=torch.float64)
    BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float64)
    AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float64)
    U = G @ w @ G.t()                                       # [cout, cin, 4, 4]
    for (co, ci) in ((0, 0), (70, 9), (127, 31)):
        got = pk[co // 64, ci // 8, ci % 8, co % 64].double().reshape(4, 4)
        assert torch.allclose(got, U[co, ci], atol=1e-6)
    d = torch.randn(cin, 4, 4, generator=g, dtype=torch.float64)     # one 4x4 input patch per channel
    V = BT @ d @ BT.t()
    M = (U[5] * V).sum(0)
    Y = AT @ M @ AT.t()
    ref = torch.nn.functional.conv2d(d[None], w[5:6])[0, 0]         # valid 3x3 correlation -> 2x2
    assert torch.allclose(Y, ref, atol=1e-10)
    # row form: wave xr computes row xr of B^T d as d[ra] + sigma * d[rb]
    rows = {0: (0, 2, -1.0), 1: (1, 2, 1.0), 2: (2, 1, -1.0), 3: (1, 3, -1.0)}
    for xr, (ra, rb, sg) in rows.items():
        t = d[:, ra] + sg * d[:, rb]                                # [cin, 4]
        v = torch.stack((t[:, 0] - t[:, 2], t[:, 1] + t[:, 2], t[:, 2] - t[:, 1], t[:, 1] - t[:, 3]), 1)
        assert torch.allclose(v, V[:, xr], atol=1e-12)
    # epilogue: s = M A per row, then Y0 = s0+s1+s2, Y1 = s1-s2-s3
    s = M @ AT.t()
    assert torch.allclose(torch.stack((s[0] + s[1] + s[2], s[1] - s[2] - s[3])), Y, atol=1e-12)


def test_tail_weight_pack():
    from deepinv_amd.hip.drunet import pack_tail_weight

    w = torch.arange(2 * 16 * 9, dtype=torch.float32).reshape(2, 16, 3, 3)
    pk = pack_tail_weight(w)
    assert pk.shape == (2, 9, 2, 8)
    for (co, ci, ky, kx) in ((0, 0, 0, 0), (1, 9, 2, 1), (1, 15, 1, 2)):
        assert pk[ci // 8, ky * 3 + kx, co, ci % 8] == w[co, ci, ky, kx]
    with pytest.raises(ValueError):
        pack_tail_weight(torch.zeros(5, 16, 3, 3))


def test_split2d_weight_pack_layout_and_products():
    """pack_split2d_weight: w = hi + lo + e with bf16 parts and |e| <= 2^-16 |w|, packed [Cout/64][Cin/16][dy][plane][dx]
    [cblk][row 64][ci 8] with the rows of each 32-row tile permuted (row 8g + 4h + e <- cout 16(g>>1) + 8h + 4(g&1) + e);
    and the numbers behind the three-product form: hi*hi alone is ~2^-9, the three leading products ~2^-17."""
    from deepinv_amd.hip.drunet import pack_split2d_weight, split2d_row_perm

    g = torch.Generator().manual_seed(3)
    cout, cin = 128, 32
    w = torch.randn(cout, cin, 3, 3, generator=g)
    pk = pack_split2d_weight(w)
    assert pk.shape == (cout // 64, cin // 16, 3, 2, 3, 2, 64, 8) and pk.dtype == torch.bfloat16
    perm = split2d_row_perm()
    assert sorted(perm.tolist()) == list(range(32))
    rec = pk.float().sum(3)                                           # hi + lo: [ct, s, dy, dx, cblk, row, ci]
    for (ct, s_, dy, dx, cb, row, ci) in ((1, 1, 1, 2, 1, 37, 3), (0, 0, 0, 0, 0, 0, 0), (1, 0, 2, 1, 0, 63, 7)):
        co = 64 * ct + 32 * (row // 32) + int(perm[row % 32])
        ref = w[co, 16 * s_ + 8 * cb + ci, dy, dx]
        assert abs(float(rec[ct, s_, dy, dx, cb, row, ci] - ref)) <= 2.0 ** -16 * abs(float(ref))
    # lane half h of the MFMA D fragment holds rows 8g + 4h .. + 3 in register quad g: with the permutation, quads 2k and
    # 2k + 1 of half h are couts 8(2k + h) .. 8(2k + h) + 7, one complete channel block
    for h in range(2):
        for k in range(2):
            rows = [8 * gq + 4 * h + e for gq in (2 * k, 2 * k + 1) for e in range(4)]
            assert [int(perm[r]) for r in rows] == list(range(8 * (2 * k + h), 8 * (2 * k + h) + 8))

    def split(x, n):
        parts, r = [], x.clone()
        for _ in range(n):
            p = r.bfloat16().float()
            parts.append(p)
            r = r - p
        return parts

    a, b = torch.randn(64, 576, generator=g), torch.randn(576, 64, generator=g)
    exact = a.double() @ b.double()
    rel = lambda m: float((m.double() - exact).norm() / exact.norm())
    a2, b2 = split(a, 2), split(b, 2)
    three = a2[1] @ b2[0] + a2[0] @ b2[1] + a2[0] @ b2[0]
    assert rel(three) < 2e-5 and rel(a2[0] @ b2[0]) > 1e-3


def test_wsplit_weight_pack_layout_and_algebra():
    """pack_wsplit_weight: [Cout/64][Cin/16][dy 3][point 4][m 2][plane 2][cblk 2][row 32][ci 8] holds the two bf16 parts of the
    F(2,3) weights U0 = g0, U1 = (g0+g1+g2)/2, U2 = (g0-g1+g2)/2, U3 = g2 of kernel row dy (rows of a 32-row tile permuted
    like pack_split2d_weight), and the identity csrc/drunet_wsplit.hip relies on reproduces the 3x3 correlation (fp64):
    V0 = d0-d2, V1 = d1+d2, V2 = d2-d1, V3 = d1-d3;  y(2j) = M0+M1+M2, y(2j+1) = M1-M2-M3 with M_k = sum_dy U_k[dy] V_k[row+dy]."""
    from deepinv_amd.hip.drunet import pack_wsplit_weight, split2d_row_perm

    g = torch.Generator().manual_seed(5)
    cout, cin = 128, 32
    w = torch.randn(cout, cin, 3, 3, generator=g)
    pk = pack_wsplit_weight(w)
    assert pk.shape == (cout // 64, cin // 16, 3, 4, 2, 2, 2, 32, 8) and pk.dtype == torch.bfloat16
    gd = w.double()
    U = torch.stack((gd[..., 0], (gd[..., 0] + gd[..., 1] + gd[..., 2]) / 2, (gd[..., 0] - gd[..., 1] + gd[..., 2]) / 2, gd[..., 2]))
    perm = split2d_row_perm()
    for (co, ci, dy, k) in ((0, 0, 0, 0), (70, 9, 2, 1), (127, 31, 1, 2), (33, 16, 0, 3)):
        ct, m, r = co // 64, (co % 64) // 32, co % 32
        row = int((perm == r).nonzero()[0, 0])          # MFMA row that carries cout r of its 32-row tile
        s, cb, c8 = ci // 16, (ci % 16) // 8, ci % 8
        hi, lo = pk[ct, s, dy, k, m, 0, cb, row, c8].double(), pk[ct, s, dy, k, m, 1, cb, row, c8].double()
        u = U[k, co, ci, dy]
        assert abs(float(hi + lo - u)) <= 2.0 ** -16 * abs(float(u)) + 1e-30
    # the algebra on one row of pairs (valid correlation of a 3-row strip with an even number of output columns)
    Wd = 10
    d = torch.randn(cin, 3, Wd + 2, generator=g, dtype=torch.float64)
    ref = torch.nn.functional.conv2d(d[None], gd[7:8])[0, 0, 0]                 # [Wd]
    d0, d1, d2, d3 = d[..., 0:Wd:2], d[..., 1:Wd + 1:2], d[..., 2:Wd + 2:2], d[..., 3:Wd + 3:2]
    V = (d0 - d2, d1 + d2, d2 - d1, d1 - d3)                                       # [cin, 3, Wd/2] each
    M = [(U[k, 7][:, :, None] * V[k]).sum((0, 1)) for k in range(4)]
    y = torch.stack((M[0] + M[1] + M[2], M[1] - M[2] - M[3]), -1).reshape(-1)
    assert torch.allclose(y, ref, atol=1e-10)
    with pytest.raises(ValueError):
        pack_wsplit_weight(torch.zeros(64, 24, 3, 3))


def test_res_block_dispatch_even_and_odd_widths(monkeypatch):
    """bf16-split precision: the Winograd operand-split kernel takes the ResBlock convolutions wherever the level's width is
    even, the direct operand-split kernel (pre-split temporary) otherwise; without the packs, the fp32 kernels (no GPU needed:
    the launch wrappers are replaced by recorders)"""
    import types
    from deepinv_amd.models import drunet as D

    calls = []
    monkeypatch.setattr(D.K, "conv3x3_wsplit", lambda g, x, w, ci, co, y, res1=None, relu=False: calls.append(("wsplit", relu, res1 is not None)))
    monkeypatch.setattr(D.K, "conv3x3_split", lambda g, x, w, ci, co, y, res1=None, relu=False, x_presplit=False, y_presplit=False,
                        gate=False: calls.append(("split", relu, res1 is not None, x_presplit, y_presplit)))
    model = D.DRUNet.__new__(D.DRUNet)
    model._conv_fp32 = lambda g, pk, x, y, relu=False, res1=None: calls.append(("fp32", relu, res1 is not None))
    pk = ((None, 64, 64), None, None, "s2d", "wsp")
    x = t = y = object()
    model._res_block(types.SimpleNamespace(width=40), pk, pk, x, t, y)
    assert calls == [("wsplit", True, False), ("wsplit", False, True)]
    calls.clear()
    model._res_block(types.SimpleNamespace(width=5), pk, pk, x, t, y)
    assert calls == [("split", True, False, False, True), ("split", False, True, True, False)]
    calls.clear()
    none = ((None, 64, 64), None, None, None, None)
    model._res_block(types.SimpleNamespace(width=40), none, none, x, t, y)
    assert calls == [("fp32", True, False), ("fp32", False, True)]


def test_sigma_operand_forms_and_shape_checks():
    """DRUNet.forward hands the noise level to the pack kernel as it came (float / per-sample values / map) instead of building
    the concatenated noise-map channel of drunet.py:226-249; the shape checks are the reference's"""
    import deepinv_amd as dinv

    m = dinv.models.DRUNet(2, 2, nc=(8, 8, 8, 8), nb=1, pretrained=None)
    x = torch.zeros(3, 2, 32, 40)
    assert m._sigma_operand(x, 0.1) == pytest.approx(0.1) and isinstance(m._sigma_operand(x, 0.1), float)
    assert isinstance(m._sigma_operand(x, torch.tensor(0.2)), float)                      # 0-dim host tensor: a scalar
    assert isinstance(m._sigma_operand(x, torch.tensor([0.2])), float)
    per = torch.tensor([0.1, 0.2, 0.3])
    assert m._sigma_operand(x, per) is per and m._sigma_operand(x, per.view(3, 1, 1, 1)).shape == (3, 1, 1, 1)
    smap = torch.rand(3, 1, 32, 40)
    assert m._sigma_operand(x, smap) is smap
    for bad in (torch.rand(2), torch.rand(3, 1, 32, 41), torch.rand(3, 2, 32, 40)):
        with pytest.raises(ValueError, match="Incorrect shape"):
            m._sigma_operand(x, bad)
    # the same map the reference would have concatenated
    assert torch.equal(m._noise_map(x, per)[:, 0, 0, 0], per) and m._noise_map(x, 0.1).shape == (3, 1, 32, 40)


def test_bf16x3_pack_carries_all_24_bits():
    """pack_down_bf16x3_weight: hi + mid + lo reproduces the fp32 weight to its last bit (the six-product stride-2 convolution of
    the fp32 setting multiplies full fp32 operands)"""
    from deepinv_amd.hip.drunet import pack_down_bf16x3_weight

    w = torch.randn(64, 32, 2, 2, generator=torch.Generator().manual_seed(3))
    p = pack_down_bf16x3_weight(w)                      # dy, dx, s, plane, cblk, co, ci
    assert p.shape == (2, 2, 2, 3, 2, 64, 8) and p.dtype == torch.bfloat16
    back = (p[:, :, :, 0].double() + p[:, :, :, 1].double() + p[:, :, :, 2].double()).permute(4, 2, 3, 5, 0, 1).reshape(64, 32, 2, 2)
    assert float((back - w.double()).abs().max()) <= float(w.abs().max()) * 2.0 ** -24
    with pytest.raises(ValueError):
        pack_down_bf16x3_weight(torch.zeros(64, 24, 2, 2))
