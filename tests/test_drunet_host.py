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


def test_bf16_split_weight_pack_is_exact_and_laid_out():
    """pack_bf16x3_weight: w = p1 + p2 + p3 with bf16 parts (error below 2^-24 |w|: the split loses nothing an fp32
    product would keep), packed [Cout/64][Cin/8][plane 3][tap 9][co 64][8]; and the six leading products reproduce
    an fp32 dot product to fp32 accuracy while the three-product form is ~2^-17 (the numbers behind drunet_bf16.hip)."""
    from deepinv_amd.hip.drunet import pack_bf16x3_weight

    g = torch.Generator().manual_seed(3)
    cout, cin = 128, 24
    w = torch.randn(cout, cin, 3, 3, generator=g)
    pk = pack_bf16x3_weight(w)
    assert pk.shape == (cout // 64, cin // 8, 3, 9, 64, 8) and pk.dtype == torch.bfloat16
    rec = pk.float().sum(2)                                           # [ct, cb, tap, co, ci]
    ref = w.reshape(cout // 64, 64, cin // 8, 8, 9).permute(0, 2, 4, 1, 3)
    assert float((rec - ref).abs().max()) <= 2.0 ** -23 * float(ref.abs().max())
    assert pk[1, 2, 0, 5, 7, 3] == w[64 + 7, 16 + 3, 1, 2].bfloat16()  # tap 5 = (ky 1, kx 2)

    def split(x, n):
        parts, r = [], x.clone()
        for _ in range(n):
            p = r.bfloat16().float()
            parts.append(p)
            r = r - p
        return parts

    a, b = torch.randn(64, 576, generator=g), torch.randn(576, 64, generator=g) / 24
    exact = a.double() @ b.double()
    rel = lambda o: float((o.double() - exact).norm() / exact.norm())
    a3, b3 = split(a, 3), split(b, 3)
    six = sum(a3[i] @ b3[j] for i, j in ((2, 0), (1, 1), (0, 2), (1, 0), (0, 1), (0, 0)))
    a2, b2 = split(a, 2), split(b, 2)
    three = a2[1] @ b2[0] + a2[0] @ b2[1] + a2[0] @ b2[0]
    assert rel(six) < 5e-7 and rel(three) < 2e-5 and rel(a2[0] @ b2[0]) > 1e-3
