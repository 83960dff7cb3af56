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
