"""DRUNet MFMA path vs the CPU oracle (same seeded weights via state_dict)."""
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
    b = model.forward_unet_torch(xin)
    assert rel_err(a, b) < 1e-4


def test_drunet_unsafe_shapes(dev):
    import deepinv_amd as dinv

    model = dinv.models.DRUNet(1, 1, pretrained=None).to(dev).eval()
    with torch.no_grad():
        y = model(torch.rand(1, 1, 37, 41, device=dev), 0.1)    # test_pad branch
        z = model(torch.rand(1, 1, 70, 100, device=dev), 0.1)   # test_onesplit branch
    assert y.shape == (1, 1, 37, 41) and z.shape == (1, 1, 70, 100)
    assert torch.isfinite(y).all() and torch.isfinite(z).all()
