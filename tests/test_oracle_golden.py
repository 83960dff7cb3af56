"""Pin the CPU oracle against golden vectors produced by the REAL reference (tests/golden/make_golden.py)
and against the reference's literal doctest vectors.  Runs on CPU (no GPU, no /root/reference needed)."""
import os

import numpy as np
import pytest
import torch

from oracle import drunet_cpu as OD
from oracle import naive
from oracle import optim_cpu as OO
from oracle import physics_cpu as O

G = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    d = np.load(os.path.join(G, name + ".npz"))
    return {k: torch.from_numpy(d[k]) for k in d.files}


def cplx(t):
    return torch.view_as_complex(t.contiguous())


def close(a, b, tol=2e-6):
    a, b = a.double(), b.double()
    a, b = a.detach(), b.detach()
    return float((a - b).norm() / b.norm().clamp_min(1e-30)) < tol


def test_mri_single_coil_2d_3d():
    d = load("mri_2d")
    assert close(O.mri_A(d["x"], d["mask"]), d["y"])
    assert close(O.mri_AT(d["y"], d["mask"]), d["xadj"])
    assert close(O.mri_prox_l2(d["z"], d["y"], 0.7, d["mask"]), d["prox"])
    assert close(O.mri_dagger(d["y"], d["mask"]), d["dagger"])
    d = load("mri_3d")
    assert close(O.mri_A(d["x"], d["mask"], True), d["y"])
    assert close(O.mri_AT(d["y"], d["mask"], True), d["xadj"])


def test_mri_multicoil_and_naive():
    d = load("multicoil_2d")
    maps = cplx(d["maps"])
    assert close(O.multicoil_A(d["x"], maps, d["mask"]), d["y"])
    assert close(O.multicoil_AT(d["y"], maps, d["mask"]), d["xadj"])
    assert close(O.multicoil_AT_rss(d["y"], d["mask"]), d["rss"])
    # fp64 direct-sum definitions agree with the reference too
    m = O.check_mask(d["mask"]).numpy()
    assert close(torch.from_numpy(naive.multicoil_A(d["x"].numpy(), maps.numpy(), m)), d["y"], 1e-5)
    assert close(torch.from_numpy(naive.multicoil_AT(d["y"].numpy(), maps.numpy(), m)), d["xadj"], 1e-5)
    d = load("multicoil_3d")
    maps = cplx(d["maps"])
    assert close(O.multicoil_A(d["x"], maps, d["mask"], True), d["y"])
    assert close(O.multicoil_AT(d["y"], maps, d["mask"], True), d["xadj"])
    m = O.check_mask(d["mask"], True).numpy()
    assert close(torch.from_numpy(naive.multicoil_A(d["x"].numpy(), maps.numpy(), m, ndim=3)), d["y"], 1e-5)


def test_mri_doctest_literal():
    """deepinv/physics/mri.py:52-76"""
    d = load("mri_doctest")
    y = O.mri_A(d["x"], d["mask"])
    lit = torch.tensor([[[[0.0000, -1.4290], [0.4564, -0.0000]], [[0.0000, 1.8622], [0.0603, -0.0000]]]])
    assert torch.allclose(y, lit, atol=1e-4)
    assert torch.equal(y, d["y"])
    # zero pattern == mask zero pattern (test_physics.py:1052-1077)
    assert torch.equal(y == 0, O.check_mask(d["mask"]).expand_as(y) == 0)


def test_mri_fft_matches_fastmri_reference():
    """test_physics.py:1575-1648: centred fft == fastMRI fft2c, re-stated independently in fp64"""
    d = load("mri_fft")
    assert close(O.im_to_kspace(d["x"]), d["k"], 1e-7)
    assert close(O.kspace_to_im(d["x"]), d["back"], 1e-7)
    xc = d["x"][:, 0].numpy() + 1j * d["x"][:, 1].numpy()
    k = naive.centered_dftn(xc, 2)
    assert close(torch.from_numpy(np.stack([k.real, k.imag], 1)), d["k"], 1e-5)


@pytest.mark.parametrize("circle", [0, 1])
def test_tomography(circle):
    d = load(f"tomo_16_circle{circle}")
    ang = d["angles"]
    assert close(O.radon_forward(d["x"], ang, bool(circle)), d["y"], 1e-5)
    assert close(O.radon_adjoint(d["v"], ang, 16, bool(circle)), d["vadj"], 1e-5)
    assert close(O.ramp_filter(d["y"]), d["ramp"], 1e-6)
    assert close(O.tomography_fbp(d["y"], ang, 16, circle=bool(circle)), d["fbp"], 1e-5)
    y_n = naive.radon_forward(d["x"][0, 0].numpy().astype(np.float64), ang.numpy(), bool(circle))
    assert close(torch.from_numpy(y_n), d["y"][0, 0], 1e-5)


def test_tomography_doctest_literal():
    """deepinv/physics/tomography.py:91-114 (values printed to 4 decimals, normalize=False)"""
    d = load("tomo_doctest")
    assert close(O.radon_forward(d["x"], d["angles"], circle=True) / d["operator_norm"], d["y"], 1e-5)
    torch.manual_seed(0)
    x = torch.randn(1, 1, 4, 4)
    y = O.radon_forward(x, torch.linspace(0, 45, steps=3), circle=True)
    lit = torch.tensor([[[[0.0000, -0.1791, -0.1719], [-0.5713, -0.4521, -0.5177], [0.0340, 0.1448, 0.2334],
                          [0.0000, -0.0448, -0.0430]]]])
    assert torch.allclose(y, lit, atol=1e-4)
    y3 = O.radon_forward(x, torch.linspace(0, 180, steps=4)[:-1], circle=True)
    lit3 = torch.tensor([[[[0.0000, -0.1806, 0.0500], [-0.5713, -0.6076, -0.6815], [0.0340, 0.3175, 0.0167],
                           [0.0000, -0.0452, 0.0989]]]])
    assert torch.allclose(y3, lit3, atol=1e-4)


@pytest.mark.parametrize("pad", ["valid", "circular", "reflect", "replicate", "constant"])
def test_blur_paddings(pad):
    d = load("blur_paddings")
    assert close(O.conv2d(d["x"], d["k"], pad), d[f"y_{pad}"])
    assert close(O.conv_transpose2d(d[f"v_{pad}"], d["k"], pad, 17, 19), d[f"vadj_{pad}"], 1e-5)


def test_blurfft_and_downsampling():
    d = load("blurfft")
    mask, angle = O.blurfft_params((3, 17, 19), d["k"])
    assert close(mask, d["mask"]) and close(torch.view_as_real(angle), d["angle"], 1e-5)
    assert close(O.blurfft_A(d["x"], mask, angle, (3, 17, 19)), d["y"])
    assert close(O.blurfft_AT(d["y"], mask, angle, (3, 17, 19)), d["xadj"])
    assert close(O.blurfft_prox_l2(d["z"], d["y"], 1.3, mask, angle, (3, 17, 19)), d["prox"])
    d = load("downsampling")
    assert close(O.downsampling_A(d["x"], d["k"], 4), d["y"])
    assert close(O.downsampling_AT(d["y"], d["k"], 4, (3, 32, 24)), d["yadj"], 1e-5)
    assert close(O.downsampling_prox_l2(d["z"], d["y"], 0.8, d["k"], 4, (3, 32, 24)), d["prox"], 1e-5)


def test_drunet():
    d = load("drunet_2ch")
    sd = OD.init_state_dict(2, 2, seed=123)
    assert torch.equal(sd["m_head.weight"], d["w_head"]), "seeded weight init is not reproducible on this box"
    with torch.no_grad():
        assert close(OD.drunet(sd, d["x"], 0.05), d["y"], 1e-5)


def test_pnp_loops():
    d = load("pnp_mri")
    maps = cplx(d["maps"])
    sd = OD.init_state_dict(2, 2, seed=123)
    A = lambda v: O.multicoil_A(v, maps, d["mask"])
    AT = lambda v: O.multicoil_AT(v, maps, d["mask"])
    den = lambda u, s: OD.drunet(sd, u, s)
    with torch.no_grad():
        r = OO.pnp_pgd(d["y"], A, AT, den, stepsize=1.0, sigma_denoiser=0.05, max_iter=3)
        assert close(r, d["rec_pgd"], 1e-5)
        prox = lambda z, y, gam: OO.prox_l2_cg(z, y, gam, A, AT, max_iter=50, tol=1e-4)
        r = OO.pnp_hqs(d["y"], prox, den, [2.0, 1.0, 0.5], [0.1, 0.05, 0.02], max_iter=3, x0=AT(d["y"]))
        assert close(r, d["rec_hqs"], 1e-4)


def test_cfg1_blurfft_pgd_plumbing():
    """BASELINE config[0]: BlurFFT 9x9 Gaussian on 1x3x256x256, 20-iter PGD, identity denoiser (CPU)."""
    d = load("cfg1_blurfft_pgd")
    x = torch.rand(1, 3, 256, 256, generator=torch.Generator().manual_seed(int(d["seed"])))
    assert torch.equal(x[..., :64, :64], d["x_crop"])
    # the golden script builds the PSF with the reference's gaussian_blur; restate it here
    ax = torch.linspace(-4, 4, 9)
    gk = torch.exp(-0.5 * ax ** 2 / 4.0)
    k = torch.outer(gk, gk)
    k = (k / k.sum())[None, None]
    mask, angle = O.blurfft_params((3, 256, 256), k)
    A = lambda v: O.blurfft_A(v, mask, angle, (3, 256, 256))
    AT = lambda v: O.blurfft_AT(v, mask, angle, (3, 256, 256))
    y = A(x)
    assert close(y[..., :64, :64], d["y_crop"], 1e-5)
    r = OO.pnp_pgd(y, A, AT, lambda u, s: u, stepsize=1.0, max_iter=20)
    assert close(r[..., :64, :64], d["rec_crop"], 1e-5)
    assert abs(float(r.double().sum()) - float(d["rec_sum"])) / abs(float(d["rec_sum"])) < 1e-5


# ------------------------------------------------------------------ round-2 fixtures (tests/golden/make_golden_r2.py)
def test_iradon_applyradon_branch():
    """Tomography(adjoint_via_backprop=False): ApplyRadon / IRadon (radon.py:396-531)"""
    d = load("tomo_applyradon")
    for c in (0, 1):
        assert close(O.radon_forward(d["x"], d["angles"], bool(c)), d[f"y_c{c}"])
        assert close(O.iradon_backproject(d[f"v_c{c}"], d["angles"], 16, bool(c)), d[f"vadj_c{c}"])
        fbp = O.iradon_backproject(O.ramp_filter(d[f"y_c{c}"]), d["angles"], 16, bool(c)) * torch.pi / (2 * 12)
        assert close(fbp, d[f"fbp_c{c}"])


def test_tomography_normalised():
    d = load("tomo_normalized")
    nrm = d["operator_norm"]
    assert close(O.radon_forward(d["x"], d["angles"]) / nrm, d["y"])
    assert close(O.radon_adjoint(d["v"], d["angles"], 16) / nrm, d["vadj"])
    assert close(O.tomography_fbp(d["y"], d["angles"], 16, operator_norm=nrm), d["fbp"])
    # the stored norm is the largest singular value of the unnormalised operator (power method, 1e-3 class)
    A = lambda v: O.radon_forward(v, d["angles"])
    AT = lambda v: O.radon_adjoint(v, d["angles"], 16)
    v = torch.randn(1, 1, 16, 16, generator=torch.Generator().manual_seed(0))
    for _ in range(60):
        v = AT(A(v))
        lam = v.norm()
        v = v / lam
    assert abs(float(lam.sqrt()) - float(nrm)) / float(nrm) < 2e-3


def test_drunet_with_unit_gain_resblocks():
    d = load("drunet_gain1")
    sd = OD.init_state_dict(2, 2, seed=321, res_gain=1.0)
    assert torch.equal(sd["m_body.0.res.0.weight"][:4, :4], d["w_body"])
    with torch.no_grad():
        assert close(OD.drunet(sd, d["x"], 0.05), d["y"], 1e-6)


def test_diffpir_matches_reference_sample_path():
    """DiffPIR (diffusion.py:289-513): schedule and a 6-step sample path with the reference's recorded noise"""
    d = load("diffpir")
    S = OO.diffpir_schedule(torch.tensor(0.05), 6, 7.0)
    assert torch.equal(S["seq"], d["seq"]) and close(S["rhos"], d["rhos"], 1e-7) and close(S["sigmas"], d["sigmas"], 1e-7)
    for tag in ("a", "b"):
        s = load("diffpir_schedule_" + tag)
        S = OO.diffpir_schedule(float(s["sigma"]), int(s["max_iter"]), float(s["lambda_"]))
        assert torch.equal(S["seq"], s["seq"]) and close(S["rhos"], s["rhos"], 1e-6) and close(S["sigmas"], s["sigmas"], 1e-7)
    sd = OD.init_state_dict(3, 3, seed=5)
    img = (3, 32, 32)
    with torch.no_grad():
        out = OO.diffpir(d["y"], lambda v: O.downsampling_AT(v, d["k"], 4, img),
                         lambda z, yy, gam: O.downsampling_prox_l2(z, yy, gam, d["k"], 4, img),
                         lambda u, s: OD.drunet(sd, u, s), list(d["draws"]), sigma=0.05, max_iter=6, noise_sigma=0.05)
    # not 2e-6 like the other fixtures: at step 0 the prox runs with gamma = 1/(2 rho) = 7e5 and its closed form
    # (z_hat - r) * gamma subtracts two nearly equal images, so the 2e-8 rounding difference between this oracle's
    # A^T (autograd transpose of conv2d) and the reference's (F.conv_transpose2d + border fold-back) is amplified to
    # 2e-3 in that step's output and decays to 2e-5 in the final sample (measured).  Any implementation that is not
    # bit-identical to the reference's A^T sees this; the bound below is the north_star's 1e-4.
    assert close(out, d["out"], 1e-4)


def test_downsampling_prox_forms():
    """Downsampling.prox_l2 (blur.py:331-363): the reference's closed form and the residual form the product evaluates
    are the same minimiser (fp64: 1e-11), but in fp32 the reference's form loses digits in proportion to gamma: at
    DiffPIR's first-step gamma = 1/(2 rho) = 7e5 it is ~5e-3 off the exact result, the residual form ~3e-7.  This is the
    measured replacement for the prose that used to justify a 1e-3 DiffPIR tolerance."""
    d = load("diffpir")
    img, f = (3, 32, 32), 4
    z = torch.rand(2, *img, generator=torch.Generator().manual_seed(0))
    for gam, ref_lo, ref_hi in ((0.7, 0.0, 1e-6), (7e5, 1e-4, 1e-1)):
        exact = O.downsampling_prox_l2(z.double(), d["y"].double(), gam, d["k"].double(), f, img)
        assert close(O.downsampling_prox_l2_residual(z.double(), d["y"].double(), gam, d["k"].double(), f, img), exact, 1e-9)
        e_ref = float((O.downsampling_prox_l2(z, d["y"], gam, d["k"], f, img).double() - exact).norm() / exact.norm())
        e_res = float((O.downsampling_prox_l2_residual(z, d["y"], gam, d["k"], f, img).double() - exact).norm() / exact.norm())
        assert ref_lo <= e_ref <= ref_hi, (gam, e_ref)
        assert e_res < 2e-6, (gam, e_res)


def test_tomography_fan_beam():
    """fan_beam_grid / Radon(fan_beam=True) / Tomography.fbp with the fan branch (radon.py:16-52, tomography.py:229-350)"""
    import numpy as np
    raw = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tomo_fan.npz"))
    fan = dict(zip([str(k) for k in raw["fan_keys"]], [float(v) for v in raw["fan"]]))
    fan["n_detector_pixels"] = int(fan["n_detector_pixels"])
    t = lambda k: torch.from_numpy(raw[k])
    for c in (0, 1):
        assert close(O.radon_fan_forward(t("x"), t("angles"), fan, bool(c)), t(f"y_c{c}"))
        assert close(O.radon_fan_adjoint(t(f"v_c{c}"), t("angles"), 16, fan, bool(c)), t(f"vadj_c{c}"))
        assert close(O.tomography_fan_fbp(t(f"y_c{c}"), t("angles"), 16, fan, None, bool(c)), t(f"fbp_c{c}"))
    a6 = torch.linspace(0, 180, 7)[:-1]                      # the reference's default fan parameters
    assert close(O.radon_fan_forward(t("x"), a6), t("y_default"))
    assert close(O.radon_fan_adjoint(t("v_default"), a6, 16), t("vadj_default"))


def test_drunet_3d_forward_and_gradients():
    """oracle forward_unet_nd (dim = 3) with the weights the reference's own initialisation produces from the stored
    seed, against the reference's output and autograd gradients (drunet.py:39-263 with Conv3d / ConvTranspose3d)"""
    import numpy as np
    raw = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "drunet3d.npz"))
    import deepinv_amd as dinv       # host-side module construction only (nn.Conv3d parameters on the CPU)

    torch.manual_seed(7)
    model = dinv.models.DRUNet(2, 2, nc=(16, 32, 64, 128), nb=1, pretrained=None, dim=3)
    assert [n for n, _ in model.named_parameters()] == [str(n) for n in raw["names"]]
    sd = {n: p.detach().clone().requires_grad_(True) for n, p in model.named_parameters()}
    x = torch.from_numpy(raw["x"]).requires_grad_(True)
    sig = torch.from_numpy(raw["sigma"]).requires_grad_(True)
    y = OD.forward_unet_nd(sd, torch.cat((x, sig), 1), nb=1, dim=3)
    assert close(y, torch.from_numpy(raw["y"]), 1e-5)
    (y * torch.from_numpy(raw["v"])).sum().backward()
    assert close(x.grad, torch.from_numpy(raw["gx"]), 1e-5) and close(sig.grad, torch.from_numpy(raw["gsigma"]), 1e-5)
    for i, n in enumerate(raw["names"]):
        n = str(n)
        assert abs(float(sd[n].grad.norm()) - float(raw["gw_norms"][i])) <= 1e-4 * float(raw["gw_norms"][i]), n
        if "gw_" + n in raw.files:
            assert close(sd[n].grad, torch.from_numpy(raw["gw_" + n]), 1e-4), n
