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
