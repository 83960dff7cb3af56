"""The drop-in claim, tested against the REAL deepinv (build container only: marker `reference`).

The product's operator objects (deepinv_amd.physics.*) are handed to the reference's own code - its optimizers
(deepinv.optim.PGD / HQS, deepinv.unfolded.unfolded_builder) and the bodies of its own property tests
(deepinv/tests/test_physics.py: adjointness :717-735, operator norm :882-927, pseudo-inverse :946-968, Blur == BlurFFT
:1338-1381) with `deepinv.physics.{MRI, MultiCoilMRI, Tomography, Blur, BlurFFT, Downsampling}` swapped for the product's
classes.  There is no GPU here: the product's ctypes binding is pointed at the host emulation of its kernel sources
(tests/emu_backend.py), so what runs is the product's Python layer + the product's kernel code, driven by the reference."""
import importlib.util
import os
import warnings

import pytest
import torch

from oracle.ref_shim import REFERENCE_ROOT, import_reference, reference_available

pytestmark = [pytest.mark.reference, pytest.mark.skipif(not reference_available(), reason="needs /root/reference")]
SWAPPED = ("MRI", "MultiCoilMRI", "Tomography", "Blur", "BlurFFT", "Downsampling")


@pytest.fixture(scope="module")
def ref():
    dinv = import_reference()
    spec = importlib.util.spec_from_file_location("ref_test_physics", os.path.join(REFERENCE_ROOT, "deepinv", "tests", "test_physics.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return dinv, mod


@pytest.fixture
def swapped(ref, monkeypatch):
    """the reference package and its physics test module with the six operator classes replaced by the product's"""
    import deepinv_amd as A
    from emu_backend import emu_backend

    dinv, mod = ref
    with emu_backend():
        for name in SWAPPED:
            monkeypatch.setattr(dinv.physics, name, getattr(A.physics, name))
        monkeypatch.setattr(mod, "MRI", A.physics.MRI)
        monkeypatch.setattr(mod, "MultiCoilMRI", A.physics.MultiCoilMRI)
        yield dinv, mod


def rng():
    return torch.Generator("cpu").manual_seed(0)


LINEAR = ["MRI", "3DMRI", "MultiCoilMRI", "3DMultiCoilMRI", "2DParallelBeamCT", "2DFanBeamCT", "fftdeblur", "deblur_valid",
          "deblur_circular", "deblur_reflect", "deblur_replicate", "deblur_constant", "super_resolution_valid",
          "super_resolution_circular", "super_resolution_reflect", "super_resolution_replicate", "super_resolution_constant"]


@pytest.mark.parametrize("name", LINEAR)
def test_reference_adjointness_body(swapped, name):
    """deepinv/tests/test_physics.py:717-735 (adjointness_test < 1e-3, and the reference's autograd adjoint_function through
    the product's A) with the product's operator behind find_operator"""
    dinv, mod = swapped
    import deepinv_amd as A
    physics = mod.find_operator(name, torch.device("cpu"))[0]
    assert type(physics).__module__.startswith("deepinv_amd."), type(physics)
    mod.test_operators_adjointness(name, torch.device("cpu"), rng())


@pytest.mark.parametrize("name", ["MRI", "MultiCoilMRI", "2DParallelBeamCT", "fftdeblur", "deblur_circular", "super_resolution_circular"])
def test_reference_norm_body(swapped, name):
    """deepinv/tests/test_physics.py:882-927: compute_sqnorm warns exactly once when it does not converge and lands within
    1e-2 of the reference value"""
    dinv, mod = swapped
    mod.test_operators_norm(name, False, torch.device("cpu"), rng())


@pytest.mark.parametrize("name", ["MRI", "MultiCoilMRI", "2DParallelBeamCT", "fftdeblur", "deblur_circular", "super_resolution_circular"])
def test_reference_pseudo_inverse_body(swapped, name):
    """deepinv/tests/test_physics.py:946-968: A_dagger(y, solver="lsqr", tol, max_iter, verbose) recovers the range component"""
    dinv, mod = swapped
    mod.test_pseudo_inverse(name, torch.device("cpu"), rng(), False)


@pytest.mark.parametrize("img_size,filter_size,filter_type", [((1, 32, 32), (1, 5, 5), "random"), ((3, 33, 33), (1, 6, 6), "directional"),
                                                              ((1, 32, 33), (1, 6, 5), "random")])
def test_reference_blur_equals_blurfft_body(swapped, img_size, filter_size, filter_type):
    """deepinv/tests/test_physics.py:1338-1381: Blur(padding="circular") and BlurFFT agree to 1e-5 (smaller images than the
    reference's 64x64: the kernels run in emulation)"""
    dinv, mod = swapped
    mod.test_blur(img_size, filter_size, filter_type, torch.device("cpu"))


def _mri_problem(A, coils=4, H=32, W=32, B=2):
    g = torch.Generator().manual_seed(3)
    x = torch.rand(B, 2, H, W, generator=g)
    maps = torch.randn(1, coils, H, W, dtype=torch.complex64, generator=g)
    maps = maps / maps.abs().pow(2).sum(dim=1, keepdim=True).sqrt()
    mask = A.utils.radial_mask(H, W, 10)
    return x, maps, mask


def test_reference_pgd_runs_over_product_multicoil_mri(ref):
    """deepinv.optim.PGD (optimizers.py:1596-1734) with the reference's own L2 / PnP(DRUNet) driving the PRODUCT's MultiCoilMRI:
    same reconstruction as with the reference's operator"""
    import deepinv_amd as A
    from emu_backend import emu_backend
    from oracle import drunet_cpu as OD

    dinv, _ = ref
    x, maps, mask = _mri_problem(A)
    den = dinv.models.DRUNet(in_channels=2, out_channels=2, pretrained=None)
    den.load_state_dict(OD.init_state_dict(2, 2, seed=5))
    den.eval()
    p_ref = dinv.physics.MultiCoilMRI(mask=mask, coil_maps=maps, img_size=(2, 32, 32), device="cpu")
    y = p_ref.A(x)
    model = dinv.optim.PGD(data_fidelity=dinv.optim.L2(), prior=dinv.optim.PnP(den), stepsize=1.0, g_param=0.05, max_iter=4,
                           early_stop=False)
    with torch.no_grad():
        rec_ref = model(y, p_ref)
        with emu_backend():
            p_amd = A.physics.MultiCoilMRI(mask=mask, coil_maps=maps, img_size=(2, 32, 32), device="cpu")
            assert float((p_amd.A(x) - y).norm() / y.norm()) < 1e-6
            rec = model(y, p_amd)
    assert float((rec - rec_ref).norm() / rec_ref.norm()) < 1e-5


def test_reference_hqs_runs_over_product_tomography(ref):
    """deepinv.optim.HQS (optimizers.py:1459-1593; prox of L2 = the PRODUCT's Tomography.prox_l2 / its CG) with a TV-free
    Tikhonov prior: same reconstruction as with the reference's operator"""
    import deepinv_amd as A
    from emu_backend import emu_backend

    dinv, _ = ref
    g = torch.Generator().manual_seed(4)
    x = torch.rand(1, 1, 16, 16, generator=g)
    p_ref = dinv.physics.Tomography(angles=12, img_width=16, circle=False, normalize=True, device="cpu")
    y = p_ref.A(x)
    model = dinv.optim.HQS(data_fidelity=dinv.optim.L2(), prior=dinv.optim.Tikhonov(), stepsize=1.0, lambda_reg=0.1, max_iter=3,
                           early_stop=False)
    with torch.no_grad():
        rec_ref = model(y, p_ref)
        with emu_backend():
            p_amd = A.physics.Tomography(angles=12, img_width=16, circle=False, normalize=True, device="cpu")
            assert abs(float(p_amd.operator_norm) - float(p_ref.operator_norm)) < 1e-4 * float(p_ref.operator_norm)
            rec = model(y, p_amd)
    assert float((rec - rec_ref).norm() / rec_ref.norm()) < 1e-4


def test_reference_unfolded_builder_trains_through_product_mri(ref):
    """deepinv.unfolded.unfolded_builder (unfolded.py:9-226) over the PRODUCT's MultiCoilMRI: loss and the gradients of the
    trainable step size / g_param equal those obtained with the reference's operator (autograd through the product's
    torch.autograd.Function: backward = the adjoint kernel)"""
    import deepinv_amd as A
    from emu_backend import emu_backend

    dinv, _ = ref
    x, maps, mask = _mri_problem(A, coils=3, H=16, W=16, B=1)

    def run(physics):
        torch.manual_seed(0)
        prior = dinv.optim.PnP(dinv.models.DnCNN(in_channels=2, out_channels=2, depth=3, nf=8, pretrained=None))
        model = dinv.unfolded.unfolded_builder("PGD", data_fidelity=dinv.optim.L2(), prior=prior,
                                               params_algo={"stepsize": 0.9, "g_param": 0.05}, max_iter=3,
                                               trainable_params=["stepsize", "g_param"])
        y = physics.A(x)
        loss = (model(y, physics) - x).pow(2).mean()
        loss.backward()
        return float(loss), {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}

    l_ref, g_ref = run(dinv.physics.MultiCoilMRI(mask=mask, coil_maps=maps, img_size=(2, 16, 16), device="cpu"))
    with emu_backend():
        l_amd, g_amd = run(A.physics.MultiCoilMRI(mask=mask, coil_maps=maps, img_size=(2, 16, 16), device="cpu"))
    assert abs(l_amd - l_ref) < 1e-5 * abs(l_ref)
    assert set(g_amd) == set(g_ref)
    for n in g_ref:
        assert float((g_amd[n] - g_ref[n]).norm() / g_ref[n].norm().clamp_min(1e-12)) < 1e-4, n


def test_product_diffpir_follows_reference_sample_path(ref):
    """deepinv.sampling.DiffPIR (diffusion.py:227-513) and the product's sampler - host-side schedule, one fused launch per
    affine update (csrc/elementwise.hip on the emulation), the product's Downsampling with its residual-form prox - draw the same
    torch.randn_like stream under one seed and must land on the same sample (denoiser: the reference's DRUNet for both)"""
    import deepinv_amd as A
    from emu_backend import emu_backend
    from oracle import drunet_cpu as OD

    dinv, _ = ref
    g = torch.Generator().manual_seed(2)
    x = torch.rand(2, 3, 32, 32, generator=g)
    den = dinv.models.DRUNet(in_channels=3, out_channels=3, pretrained=None)
    den.load_state_dict(OD.init_state_dict(3, 3, seed=7))
    den.eval()
    p_ref = dinv.physics.Downsampling(img_size=(3, 32, 32), filter="bicubic", factor=4, padding="circular", device="cpu",
                                      noise_model=dinv.physics.GaussianNoise(0.05))
    torch.manual_seed(1)
    y = p_ref(x)
    kw = dict(sigma=0.05, max_iter=8, zeta=0.1, lambda_=7.0, device="cpu")
    with torch.no_grad():
        out_ref = dinv.sampling.DiffPIR(den, dinv.optim.L2(), **kw)(y, p_ref, seed=3)
        with emu_backend():
            p_amd = A.physics.Downsampling(img_size=(3, 32, 32), filter="bicubic", factor=4, padding="circular", device="cpu",
                                           noise_model=A.physics.GaussianNoise(0.05))
            sampler = A.sampling.DiffPIR(den, A.optim.L2(), **kw)
            steps = sampler._host_schedule()
            out = sampler(y, p_amd, seed=3)
    assert len(steps) == 8 and steps[-1]["last"] and not steps[0]["last"]
    # the reference's closed-form prox loses ~5e-3 to cancellation at the first steps' gamma (tests/test_oracle_golden.py::
    # test_downsampling_prox_forms); the sample paths then contract: 5e-4 covers it, a wrong coefficient or a shifted noise
    # draw is O(1)
    assert float((out - out_ref).norm() / out_ref.norm()) < 5e-4      # measured 5e-5


@pytest.mark.parametrize("shape", [(2, 3, 37, 41), (1, 2, 70, 100), (1, 1, 65, 129), (2, 2, 9, 20, 33)])
def test_unsafe_shape_strategies_equal_the_reference(ref, shape):
    """DRUNet's handling of shapes its U-Net cannot take (deepinv/models/drunet.py:252-262): the product's replicate-padded and
    four-window evaluations (models/drunet.py: run_replicate_padded, run_four_windows - written from the description, not from
    the reference's helpers) against deepinv.models.utils.test_pad / test_onesplit on a position-dependent toy network"""
    from deepinv.models.utils import test_onesplit as ref_split
    from deepinv.models.utils import test_pad as ref_pad

    from deepinv_amd.models.drunet import run_four_windows, run_replicate_padded

    x = torch.randn(*shape, generator=rng())
    nd = len(shape) - 2
    conv = {2: torch.nn.Conv2d, 3: torch.nn.Conv3d}[nd](shape[1], 4, 5, padding=2)

    def net(v):      # translation-variant on purpose: a stitching mistake cannot hide behind shift invariance
        ramp = sum(torch.arange(v.shape[2 + i]).float().view(*[-1 if j == i else 1 for j in range(nd)]) for i in range(nd))
        return conv(v) * (1.0 + 0.01 * ramp)

    with torch.no_grad():
        assert torch.equal(run_replicate_padded(net, x, multiple=16), ref_pad(net, x, modulo=16))
        if nd == 2:
            assert torch.equal(run_four_windows(net, x, field=64), ref_split(net, x, refield=64))
            assert torch.equal(run_four_windows(net, x, field=16), ref_split(net, x, refield=16))


@pytest.mark.parametrize("padding", ["valid", "circular", "reflect", "replicate", "constant"])
@pytest.mark.parametrize("fshape", [(1, 1, 3, 3, 3), (1, 2, 2, 4, 3), (2, 1, 3, 5, 2), (1, 2, 4, 3, 4)])
def test_volume_blur_equals_the_reference(ref, padding, fshape):
    """3-D blur (deepinv/physics/functional/convolution.py:333-640, Blur on 5-D tensors blur.py:535-561): the product's conv3d,
    conv_transpose3d, their FFT forms and Blur.A / A_adjoint on volumes (its kernels on the host emulation) against the
    reference's own functions on the same inputs, every padding, broadcast and per-sample / per-channel filters, even sizes"""
    import deepinv_amd as A
    from emu_backend import emu_backend

    dinv, _ = ref
    rF = dinv.physics.functional
    g = rng()
    x = torch.randn(2, 2, 6, 9, 8, generator=g)
    k = torch.rand(*fshape, generator=g)
    y_ref = rF.conv3d(x, k, padding=padding)
    v = torch.randn(y_ref.shape, generator=g)
    try:
        xt_ref = rF.conv_transpose3d(v, k, padding=padding)
    except RuntimeError as e:       # (its border folding breaks on a size-2 axis, convolution.py:756: nothing to compare with)
        pytest.skip(f"the reference's conv_transpose3d fails on this filter shape: {e}")
    rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
    from oracle import physics_cpu as O
    assert rel(O.conv3d(x, k, padding), y_ref) < 1e-6 and rel(O.conv_transpose3d(v, k, padding, x.shape[2:]), xt_ref) < 1e-5   # pins the oracle
    with emu_backend():
        pF = A.physics.functional
        assert rel(pF.conv3d(x, k, padding=padding), y_ref) < 1e-5
        assert rel(pF.conv_transpose3d(v, k, padding=padding), xt_ref) < 1e-5
        assert rel(pF.conv3d(x, k, padding=padding, correlation=True), rF.conv3d(x, k, padding=padding, correlation=True)) < 1e-5
        assert rel(pF.conv3d_fft(x, k, padding=padding), rF.conv3d_fft(x, k, padding=padding)) < 1e-4
        assert rel(pF.conv_transpose3d_fft(v, k, padding=padding), rF.conv_transpose3d_fft(v, k, padding=padding)) < 1e-4
        for use_fft in (False, True):
            p = A.physics.Blur(filter=k, padding=padding, use_fft=use_fft, device="cpu")
            q = dinv.physics.Blur(filter=k, padding=padding, use_fft=use_fft, device="cpu")
            assert rel(p.A(x), q.A(x)) < 1e-4 and rel(p.A_adjoint(v), q.A_adjoint(v)) < 1e-4
