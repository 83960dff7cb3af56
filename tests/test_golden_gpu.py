"""HIP path vs golden vectors generated from the REAL reference (tests/golden/*.npz), tolerance 1e-4."""
import os

import numpy as np
import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
TOL = 1e-4


def load(name, dev):
    d = np.load(os.path.join(G, name + ".npz"))
    return {k: torch.from_numpy(d[k]).to(dev) for k in d.files}


def cplx(t):
    return torch.view_as_complex(t.contiguous())


def test_mri_golden(dev):
    import deepinv_amd as dinv

    d = load("mri_2d", dev)
    p = dinv.physics.MRI(mask=d["mask"], img_size=(2, 17, 11), device=dev)
    y = p.A(d["x"])
    assert rel_err(y, d["y"]) < TOL and torch.equal(y == 0, d["y"] == 0)
    assert rel_err(p.A_adjoint(d["y"]), d["xadj"]) < TOL
    assert rel_err(p.prox_l2(d["z"], d["y"], 0.7), d["prox"]) < TOL
    assert rel_err(p.A_dagger(d["y"]), d["dagger"]) < TOL
    d = load("mri_3d", dev)
    p = dinv.physics.MRI(mask=d["mask"], img_size=(2, 5, 17, 11), three_d=True, device=dev)
    assert rel_err(p.A(d["x"]), d["y"]) < TOL and rel_err(p.A_adjoint(d["y"]), d["xadj"]) < TOL
    d = load("mri_doctest", dev)
    p = dinv.physics.MRI(mask=d["mask"], device=dev)
    assert torch.allclose(p(d["x"]), d["y"], atol=1e-5)
    d = load("mri_fft", dev)
    mix = dinv.physics.MRIMixin()
    assert rel_err(mix.im_to_kspace(d["x"]), d["k"]) < 1e-6 and rel_err(mix.kspace_to_im(d["x"]), d["back"]) < 1e-6


def test_multicoil_golden(dev):
    import deepinv_amd as dinv

    for name, td, img in (("multicoil_2d", False, (2, 17, 11)), ("multicoil_3d", True, (2, 4, 16, 12))):
        d = load(name, dev)
        p = dinv.physics.MultiCoilMRI(mask=d["mask"], coil_maps=cplx(d["maps"]), img_size=img, three_d=td, device=dev)
        assert rel_err(p.A(d["x"]), d["y"]) < TOL
        assert rel_err(p.A_adjoint(d["y"]), d["xadj"]) < TOL
        if "rss" in d:
            assert rel_err(p.A_adjoint(d["y"], rss=True), d["rss"]) < TOL


@pytest.mark.parametrize("circle", [0, 1])
def test_tomography_golden(dev, circle):
    import deepinv_amd as dinv

    d = load(f"tomo_16_circle{circle}", dev)
    p = dinv.physics.Tomography(angles=d["angles"], img_width=16, circle=bool(circle), normalize=False, device=dev)
    assert rel_err(p.A(d["x"]), d["y"]) < TOL
    assert rel_err(p.A_adjoint(d["v"]), d["vadj"]) < TOL
    assert rel_err(p.filter(d["y"]), d["ramp"]) < TOL
    assert rel_err(p.A_dagger(d["y"], fbp=True), d["fbp"]) < TOL


def test_blur_golden(dev):
    import deepinv_amd as dinv

    d = load("blur_paddings", dev)
    for pad in ("valid", "circular", "reflect", "replicate", "constant"):
        p = dinv.physics.Blur(filter=d["k"], padding=pad, device=dev)
        assert rel_err(p.A(d["x"]), d[f"y_{pad}"]) < TOL
        assert rel_err(p.A_adjoint(d[f"v_{pad}"]), d[f"vadj_{pad}"]) < TOL
    d = load("blurfft", dev)
    p = dinv.physics.BlurFFT(img_size=(3, 17, 19), filter=d["k"], device=dev)
    assert rel_err(p.mask, d["mask"]) < TOL
    assert rel_err(p.A(d["x"]), d["y"]) < TOL and rel_err(p.A_adjoint(d["y"]), d["xadj"]) < TOL
    assert rel_err(p.prox_l2(d["z"], d["y"], 1.3), d["prox"]) < TOL
    d = load("downsampling", dev)
    p = dinv.physics.Downsampling(img_size=(3, 32, 24), filter="bicubic", factor=4, padding="circular", device=dev)
    assert rel_err(p.filter, d["k"]) < 1e-6
    assert rel_err(p.A(d["x"]), d["y"]) < TOL and rel_err(p.A_adjoint(d["y"]), d["yadj"]) < TOL
    assert rel_err(p.prox_l2(d["z"], d["y"], 0.8), d["prox"]) < TOL


def test_drunet_and_pnp_loops_golden(dev):
    import deepinv_amd as dinv
    from oracle import drunet_cpu as OD

    sd = OD.init_state_dict(2, 2, seed=123)
    den = dinv.models.DRUNet(2, 2, pretrained=None).to(dev).eval()
    den.load_state_dict(sd)
    d = load("drunet_2ch", dev)
    with torch.no_grad():
        assert rel_err(den(d["x"], 0.05), d["y"]) < TOL
    d = load("pnp_mri", dev)
    p = dinv.physics.MultiCoilMRI(mask=d["mask"], coil_maps=cplx(d["maps"]), img_size=(2, 32, 32), device=dev)
    m = dinv.optim.PGD(data_fidelity=dinv.optim.L2(), prior=dinv.optim.PnP(den), stepsize=1.0, g_param=0.05, max_iter=3)
    assert rel_err(m(d["y"], p), d["rec_pgd"]) < TOL
    m = dinv.optim.HQS(data_fidelity=dinv.optim.L2(), prior=dinv.optim.PnP(den), stepsize=[2.0, 1.0, 0.5],
                       g_param=[0.1, 0.05, 0.02], max_iter=3)
    assert rel_err(m(d["y"], p), d["rec_hqs"]) < TOL


def test_cfg1_blurfft_pgd_golden(dev):
    """BASELINE config[0] through the HIP path"""
    import deepinv_amd as dinv

    d = load("cfg1_blurfft_pgd", dev)
    x = torch.rand(1, 3, 256, 256, generator=torch.Generator().manual_seed(int(d["seed"]))).to(dev)
    h = dinv.physics.functional.gaussian_blur(psf_size=(9, 9), sigma=(2.0, 2.0))
    p = dinv.physics.BlurFFT(img_size=(3, 256, 256), filter=h, device=dev)
    y = p.A(x)
    assert rel_err(y[..., :64, :64], d["y_crop"]) < TOL

    class Id(torch.nn.Module):
        def forward(self, u, s):
            return u

    m = dinv.optim.PGD(data_fidelity=dinv.optim.L2(), prior=dinv.optim.PnP(Id()), stepsize=1.0, g_param=0.05, max_iter=20)
    r = m(y, p)
    assert rel_err(r[..., :64, :64], d["rec_crop"]) < TOL
    assert abs(float(r.double().sum()) - float(d["rec_sum"])) / abs(float(d["rec_sum"])) < 1e-4
