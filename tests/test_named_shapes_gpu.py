"""BASELINE.json configs[2..4] at their NAMED shapes (per-GPU shard batch) on the HIP path against outputs of the REAL
reference (tests/golden/cfg{3,4,5}_named.npz, written by tests/golden/make_golden_r3.py): operators AND loops.
Unit 0 of every batch is the fixture's seeded input; the other units are different random inputs, so the batched
kernel paths (8 images per Radon pixel group, 2 volumes, 16 images) are the ones that run.  Large reference outputs
are stored as the strided subsample flat[::stride]; the same subsample of the HIP output is compared."""
import os

import numpy as np
import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
TOL = 1e-4          # north_star: 1e-4 relative fp32


def gen(seed):
    return torch.Generator().manual_seed(seed)


def load(name):
    d = np.load(os.path.join(G, name + ".npz"))
    return {k: torch.from_numpy(np.asarray(d[k])) for k in d.files}


def sub(t, stride):
    return t.detach().reshape(-1)[::stride]


def test_cfg3_tomography_512_720(dev):
    """Tomography 512x512, 720 angles, 8 images (= 64 / 8 GPUs): operator norm, A, exact adjoint, ramp filter, FBP, and
    FBP-initialised PnP-HQS (3 iterations, prox by exactly 3 CG iterations, DRUNet 1->1) against the reference"""
    import deepinv_amd as dinv
    from oracle import drunet_cpu as OD

    d = load("cfg3_named")
    st = int(d["stride"])
    W, nang, B = 512, 720, 8
    x = torch.cat((torch.rand(1, 1, W, W, generator=gen(50)), torch.rand(B - 1, 1, W, W, generator=gen(500)))).to(dev)
    p = dinv.physics.Tomography(angles=nang, img_width=W, circle=False, normalize=True, device=dev, max_iter=3, tol=1e-30)
    assert abs(float(p.operator_norm) - float(d["operator_norm"])) < 1e-4 * float(d["operator_norm"])
    y = p.A(x)
    assert y.shape == (B, 1, 725, nang)
    assert rel_err(sub(y[:1], st), d["y"]) < TOL
    v = torch.cat((torch.randn(1, 1, 725, nang, generator=gen(51)), torch.randn(B - 1, 1, 725, nang, generator=gen(501)))).to(dev)
    assert rel_err(sub(p.A_adjoint(v)[:1], st), d["vadj"]) < TOL
    assert rel_err(sub(p.filter(y)[:1], st), d["ramp"]) < TOL
    assert rel_err(sub(p.A_dagger(y, fbp=True)[:1], st), d["fbp"]) < TOL
    den = dinv.models.DRUNet(1, 1, pretrained=None).to(dev).eval()
    den.load_state_dict(OD.init_state_dict(1, 1, seed=int(d["drunet_seed"])))
    model = dinv.optim.HQS(data_fidelity=dinv.optim.L2(), prior=dinv.optim.PnP(den), stepsize=[float(s) for s in d["steps"]],
                           g_param=[float(s) for s in d["sigs"]], max_iter=3, early_stop=False,
                           custom_init=lambda yy, pp: pp.A_dagger(yy, fbp=True))
    with torch.no_grad():
        rec = model(y, p)
    assert rel_err(sub(rec[:1], st), d["rec"]) < TOL


def test_cfg4_multicoil_3d_12x16x256x256(dev):
    """3-D MultiCoilMRI 12 coils 16x256x256, 2 volumes (= 8 / 4 GPUs): A (+ exact zero pattern), A_adjoint, and the
    unfolded PGD training step (10 iterations, DRUNet dim=3 nc=16..128 nb=1, forward + backward): reconstruction, loss and
    the gradients of the trainable step size / g_param / head weights against the reference"""
    import deepinv_amd as dinv

    d = load("cfg4_named")
    st, sty = int(d["stride"]), int(d["stride_y"])
    coils, vol, B = 12, (16, 256, 256), 2
    x = torch.cat((torch.rand(1, 2, *vol, generator=gen(60)), torch.rand(B - 1, 2, *vol, generator=gen(600)))).to(dev)
    maps = (torch.randn(1, coils, *vol, dtype=torch.complex64, generator=gen(61)) / coils ** 0.5).to(dev)
    mask = torch.zeros(*vol)
    mask[..., ::4] = 1
    mask[..., 118:138] = 1
    p = dinv.physics.MultiCoilMRI(mask=mask.to(dev), coil_maps=maps, img_size=(2, *vol), three_d=True, device=dev)
    y = p.A(x)
    assert rel_err(sub(y[:1], sty), d["y"]) < TOL
    assert bool(((y == 0) == (p.mask[:, :, None].expand_as(y) == 0)).all())       # bit-exact mask indexing
    assert rel_err(sub(p.A_adjoint(y)[:1], st), d["yadj"]) < TOL
    torch.manual_seed(int(d["drunet_seed"]))
    den = dinv.models.DRUNet(2, 2, nc=(16, 32, 64, 128), nb=1, pretrained=None, dim=3).to(dev)
    assert torch.equal(den.m_head.weight.detach().reshape(-1)[:16].cpu(), d["w_probe"])   # the reference's weights
    model = dinv.unfolded.unfolded_builder("PGD", data_fidelity=dinv.optim.L2(), prior=dinv.optim.PnP(den),
                                           params_algo={"stepsize": 1.0, "g_param": 0.05, "lambda": 1.0},
                                           max_iter=int(d["max_iter"]), trainable_params=["stepsize", "g_param"],
                                           device=dev).to(dev)
    rec = model(y[:1], p)
    loss = (rec - x[:1]).pow(2).mean()
    loss.backward()
    assert rel_err(sub(rec, st), d["rec"]) < TOL
    assert abs(float(loss.detach()) - float(d["loss"])) < TOL * float(d["loss"])
    gp = dict(model.named_parameters())
    # gradients at the north_star tolerance (measured on an MI355X: 5.1e-7, 5.9e-7, 2.1e-7 - profiles/r04_gpu_tests.log; the 1e-3
    # of round 3 was a guess about ReLU-mask flips that this shape does not show)
    errs = {}
    for name, key in (("init_params_algo.stepsize.0", "grad_stepsize"), ("init_params_algo.g_param.0", "grad_g_param")):
        g_hip, g_ref = float(gp[name].grad), float(d[key])
        errs[key] = abs(g_hip - g_ref) / abs(g_ref)
        assert errs[key] < TOL, (name, g_hip, g_ref)
    errs["grad_head"] = rel_err(den.m_head.weight.grad.reshape(-1), d["grad_head"])
    print("cfg4 named gradient errors vs the reference:", {k: f"{v:.2e}" for k, v in errs.items()})      # (pytest -s)
    assert errs["grad_head"] < TOL


def test_cfg5_downsampling_diffpir_256(dev, monkeypatch):
    """x4 super-resolution on 3x256x256, 16 images (= 128 / 8 GPUs): A, A_adjoint, prox_l2, and DiffPIR (5 steps,
    DRUNet 3->3) replaying the reference's torch.randn_like draws (regenerated from the stored seed)"""
    import deepinv_amd as dinv
    from oracle import drunet_cpu as OD

    d = load("cfg5_named")
    st = int(d["stride"])
    img, f, B = (3, 256, 256), 4, 16
    x = torch.cat((torch.rand(2, *img, generator=gen(70)), torch.rand(B - 2, *img, generator=gen(700)))).to(dev)
    p = dinv.physics.Downsampling(img_size=img, filter="bicubic", factor=f, padding="circular", device=dev,
                                  noise_model=dinv.physics.GaussianNoise(0.05))
    y = p.A(x)
    assert rel_err(y[:2], d["y"]) < TOL
    z = torch.cat((torch.rand(2, *img, generator=gen(71)), torch.rand(B - 2, *img, generator=gen(701)))).to(dev)
    assert rel_err(sub(p.A_adjoint(y)[:2], st), d["yadj"]) < TOL
    assert rel_err(sub(p.prox_l2(z, y, 0.7)[:2], st), d["prox"]) < TOL
    den = dinv.models.DRUNet(3, 3, pretrained=None).to(dev).eval()
    den.load_state_dict(OD.init_state_dict(3, 3, seed=int(d["drunet_seed"])))
    yn = (d["y"][:1] + 0.05 * torch.randn(1, 3, 64, 64, generator=gen(73))).to(dev)
    draws = gen(74)
    monkeypatch.setattr(torch, "randn_like", lambda t, **kw: torch.randn(t.shape, generator=draws).to(t.device))
    sampler = dinv.sampling.DiffPIR(den, dinv.optim.L2(), sigma=0.05, max_iter=5, zeta=0.1, lambda_=7.0, device=dev)
    assert torch.equal(sampler.seq.cpu(), d["seq"])
    out = sampler(yn, p)
    # the reference's own closed-form prox is ~5e-3 off the exact minimiser at the first step's gamma = 7e5 (fp32
    # cancellation, tests/test_oracle_golden.py::test_downsampling_prox_forms); the product's residual form is not, so the
    # distance to the reference's sample is the REFERENCE's rounding error: `out_err_vs_exact` in the fixture
    assert rel_err(sub(out, st), d["out"]) < max(TOL, 2.0 * float(d["out_err_vs_exact"]))
    assert rel_err(sub(out, st), d["out_exact"]) < TOL       # ... and against the fp64 evaluation of the same sample path


def test_cfg2_unit_gain_with_the_bf16x3_winograd_form(dev, monkeypatch):
    """The headline configuration, 50 iterations, O(1)-gain ResBlocks, with the F(4x4) launches of conv_precision = "fp32" in their
    bf16 x 3 form (hip/drunet.py: FP32_WINOGRAD4_BF16X3 - three-part operand split, six products on the bf16 matrix cores): the
    same distance to the REAL reference's reconstruction as the fp32-MFMA form (both are a rounding pattern of the same fp32
    arithmetic; printed with pytest -s), and the two forms within 1e-5 of each other after 50 denoiser calls"""
    import deepinv_amd as dinv
    from bench import make_problem
    from deepinv_amd.hip import drunet as K
    from oracle import drunet_cpu as OD

    d = load("cfg2_named")
    st, iters = int(d["stride"]), int(d["iters"])
    physics, x, y, _, _ = make_problem(dinv, 32, 0, 320, 320, 8, dev)
    den = dinv.models.DRUNet(2, 2, pretrained=None).to(dev).eval()
    den.load_state_dict(OD.init_state_dict(2, 2, seed=int(d["drunet_seed"]), res_gain=float(d["res_gain"])))
    den.conv_precision = "fp32"
    model = dinv.optim.PGD(data_fidelity=dinv.optim.L2(), prior=dinv.optim.PnP(den), stepsize=1.0, g_param=0.05,
                           max_iter=iters, early_stop=False)
    recs, calls = {}, []
    real = K.conv3x3_winograd4_bf16x3
    monkeypatch.setattr(K, "conv3x3_winograd4_bf16x3", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    for form in (False, True):
        monkeypatch.setattr(K, "FP32_WINOGRAD4_BF16X3", form)
        with torch.no_grad():
            recs[form] = model(y, physics)
        assert bool(calls) == form                      # the packs are rebuilt when the switch changes
    errs = {form: rel_err(sub(recs[form][:1], st), d["rec_gain"]) for form in recs}
    between = rel_err(recs[True], recs[False])
    print("cfg2 unit gain, 50 it, vs the reference:", {("bf16x3" if k else "fp32 mfma"): f"{v:.2e}" for k, v in errs.items()},
          f"between the forms: {between:.2e}")
    assert errs[True] < TOL and errs[True] < 1.5 * errs[False] + 1e-6
    assert between < 1e-5


def full_length_cfg3(dinv, dev, d, B=8, precisions=None, runs=1, exact_cg_counts=False, image_seed=50, images=None):
    """BASELINE configs[2] at FULL length on the per-GPU shard (8 images 512x512, 720 angles): FBP-initialised PnP-HQS, 30
    iterations, the prox by CG with the reference's DEFAULT settings (max_iter 50, tol 1e-4: the number of CG iterations is
    decided by `torch.all(residual < tol)` over the BATCH, conjugate_gradient.py:61), DRUNet(1->1), the schedules of bench.py.
    The shard holds 8 copies of the fixture's image: the batch-coupled stopping test then stops every prox where the reference's
    single-image run stopped it, and the batched kernel paths (8 images per Radon pixel group) are still the ones that run.
    exact_cg_counts: the host reads the device-side convergence flag after EVERY CG iteration (optim/linear.py: CG_CHECK_EVERY = 1;
    the iterates do not depend on it), so that the A^T A calls of each prox are exactly its CG iterations + 1.
    Returns {precision: errors of image 0 against the reference's reconstruction / denoiser outputs, CG counts, seconds}."""
    import time

    from oracle import drunet_cpu as OD

    st, stt, iters = int(d["stride"]), int(d["stride_trace"]), int(d["iters"])
    W, nang = (int(d["width"]), int(d["angles"])) if "width" in d else (512, 720)
    if images is not None:          # a batch of DISTINCT units (fixtures with one reconstruction / trace row per unit)
        x, B = images.to(dev), images.shape[0]
    else:
        x = torch.rand(1, 1, W, W, generator=gen(image_seed)).expand(B, 1, W, W).contiguous().to(dev)
    p = dinv.physics.Tomography(angles=nang, img_width=W, circle=False, normalize=True, device=dev)
    assert abs(float(p.operator_norm) - float(d["operator_norm"])) < 1e-4 * float(d["operator_norm"])
    assert (p.max_iter, p.tol) == (int(d["cg_max_iter"]), float(d["cg_tol"]))
    y = p.A(x)
    den = dinv.models.DRUNet(1, 1, pretrained=None).to(dev).eval()
    den.load_state_dict(OD.init_state_dict(1, 1, seed=int(d["drunet_seed"])))
    model = dinv.optim.HQS(data_fidelity=dinv.optim.L2(), prior=dinv.optim.PnP(den), stepsize=[float(v) for v in d["steps"]],
                           g_param=[float(v) for v in d["sigs"]], max_iter=iters, early_stop=False,
                           custom_init=lambda yy, pp: pp.A_dagger(yy, fbp=True))
    res = {}
    from deepinv_amd.models.drunet import CONV_PRECISIONS
    from deepinv_amd.optim import linear
    ata, prox, check_every = p.A_adjoint_A, p.prox_l2, linear.CG_CHECK_EVERY
    for prec in precisions or CONV_PRECISIONS:
        den.conv_precision = prec
        for _ in range(runs):
            trace, calls, per_prox = [], [0], []
            hook = den.register_forward_hook(
                (lambda m, i, o: trace.append(torch.stack([u.detach().reshape(-1)[::stt] for u in o]))) if images is not None else
                (lambda m, i, o: trace.append(o[:1].detach().reshape(-1)[::stt])))

            def counting(v, **kw):
                calls[0] += 1
                return ata(v, **kw)

            def counting_prox(*a, **kw):
                before = calls[0]
                out = prox(*a, **kw)
                per_prox.append(calls[0] - before)
                return out

            p.A_adjoint_A, p.prox_l2 = counting, counting_prox
            if exact_cg_counts:
                linear.CG_CHECK_EVERY = 1
            try:
                with torch.no_grad():
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    rec = model(y, p)
                    torch.cuda.synchronize()
                    dt = time.perf_counter() - t0
            finally:
                hook.remove()
                del p.A_adjoint_A, p.prox_l2
                linear.CG_CHECK_EVERY = check_every
        tr = torch.stack([t.cpu() for t in trace])
        trace_err = [rel_err(a, b) for a, b in zip(tr, d["den_outs"])]
        if images is not None:
            per_unit = [rel_err(sub(rec[k:k + 1], st), d["rec"][k]) for k in range(B)]
            res[prec] = {"vs_reference": max(per_unit), "per_unit": per_unit, "trace_max": max(trace_err), "trace": trace_err,
                         "ata_per_prox": per_prox, "ata_calls": calls[0], "ata_calls_reference": int(d["n_ata"].sum()), "seconds": dt,
                         "finite": bool(torch.isfinite(rec).all())}
            continue
        res[prec] = {"vs_reference": rel_err(sub(rec[:1], st), d["rec"]),
                     "trace_max": max(trace_err), "trace": trace_err, "ata_per_prox": per_prox,
                     "copies_identical": bool(torch.equal(rec[:1].expand_as(rec), rec)),
                     "ata_calls": calls[0], "ata_calls_reference": int(d["n_ata"].sum()), "seconds": dt,
                     "finite": bool(torch.isfinite(rec).all())}
    return res


def test_cfg3_fbp_pnp_hqs_full_length_30_iterations(dev):
    """configs[2] at its FULL length against the REAL reference (tests/golden/cfg3_full.npz, make_golden_r5.py: deepinv.optim.HQS,
    deepinv/optim/optimizers.py:1459-1593, prox by deepinv's CG with its default stopping rule): the reconstruction after 30
    iterations within 1e-4 in both conv_precision settings, and the denoiser output of every iteration.

    The iterates in between are NOT a 1e-4 quantity, by the reference's own construction: its prox is a CG that stops at
    `torch.all(residual < 1e-4)` after 5-23 iterations on A^T A + I / gamma with gamma up to 402 (condition number ~ gamma
    ||A||^2) - a TRUNCATED solve whose iteration count flips by one or two under fp32 rounding whenever a residual lands near
    the threshold.  Measured (profiles/r05_cfg3_cg_diag.log): the product's CG counts equal the reference's in 26 of 30 proxes;
    where they agree the denoiser outputs are within 1e-3 .. 1.7e-3 (decaying to 1e-4), the three iterations right after a
    one-step difference (21 / 23, 16 / 14, 10 / 9 CG iterations) show 1.1e-2, 9.1e-3, 6.8e-3, and HQS contracts all of it to 9.4e-5
    at the end - the same number in both conv_precision settings, i.e. not a property of the kernels' arithmetic.  Asserted:
    the end point (1e-4), the envelope of the intermediate outputs (3e-2; 3e-3 wherever the CG counts agree), at most six
    proxes with a different CG count, the same total within 5 %."""
    import deepinv_amd as dinv

    if not os.path.exists(os.path.join(G, "cfg3_full.npz")):
        pytest.skip("tests/golden/cfg3_full.npz not generated (tests/golden/make_golden_r5.py cfg3: ~1.5 h of CPU)")
    d = load("cfg3_full")
    res = full_length_cfg3(dinv, dev, d, exact_cg_counts=True)
    print("cfg3 full length:", {k: {a: (f"{b:.2e}" if isinstance(b, float) else b) for a, b in v.items() if a != "trace"} for k, v in res.items()})
    ref_counts = [int(v) for v in d["n_ata"]]
    for prec, r in res.items():
        assert r["finite"] and r["copies_identical"]
        assert r["vs_reference"] < TOL, (prec, r)
        assert r["trace_max"] < 3e-2, (prec, r)
        assert len(r["ata_per_prox"]) == len(ref_counts) == len(r["trace"])
        differ = [i for i, (a, b) in enumerate(zip(r["ata_per_prox"], ref_counts)) if a != b]
        assert len(differ) <= 6, (prec, r["ata_per_prox"], ref_counts)
        assert abs(r["ata_calls"] - r["ata_calls_reference"]) <= 0.05 * r["ata_calls_reference"]
        # iteration i's denoiser input is prox i's output: where that prox ran the reference's number of iterations
        agree = [e for i, e in enumerate(r["trace"]) if i not in differ]
        assert max(agree) < 3e-3, (prec, r["trace"], differ)


def _assert_cfg3_run(res, d):
    """A prox's CG count is where a float residual crosses the tolerance: it moves by one or two with the last bits of the iterate (the
    fp32 and bf16-split settings of ONE run disagree at a prox or two, and so do two builds of the Radon kernels), so the PROFILE of
    counts is compared: no prox off by more than 3 applications, the absolute differences summing to at most 6 % of the reference's
    total, the total within 5 %; where the counts agree the denoiser inputs agree to 3e-3."""
    ref_counts = [int(v) for v in d["n_ata"]]
    for prec, r in res.items():
        assert r["finite"]
        assert r["vs_reference"] < TOL, (prec, {k: v for k, v in r.items() if k != "trace"})
        assert r["trace_max"] < 3e-2, (prec, r["trace"])
        assert len(r["ata_per_prox"]) == len(ref_counts) == len(r["trace"])
        differ = [i for i, (a, b) in enumerate(zip(r["ata_per_prox"], ref_counts)) if a != b]
        deltas = [abs(a - b) for a, b in zip(r["ata_per_prox"], ref_counts)]
        assert max(deltas) <= 3 and sum(deltas) <= 0.06 * sum(ref_counts), (prec, r["ata_per_prox"], ref_counts)
        assert abs(r["ata_calls"] - r["ata_calls_reference"]) <= 0.05 * r["ata_calls_reference"]
        agree = [e for i, e in enumerate(r["trace"]) if i not in differ]
        assert max(agree) < 3e-3, (prec, r["trace"], differ)


def test_cfg3_full_length_second_draw(dev):
    """A SECOND full-length draw of configs[2] against the real reference (tests/golden/cfg3_full_b.npz, make_golden_r6.py: another
    image, another DRUNet initialisation): the 1e-4 end point of the 30-iteration FBP + PnP-HQS loop is not a property of one seed
    (VERDICT r5 weak #1).  Same assertions as the first draw."""
    import deepinv_amd as dinv

    if not os.path.exists(os.path.join(G, "cfg3_full_b.npz")):
        pytest.skip("tests/golden/cfg3_full_b.npz not generated (tests/golden/make_golden_r6.py cfg3_b: ~1.5 h of CPU)")
    d = load("cfg3_full_b")
    res = full_length_cfg3(dinv, dev, d, exact_cg_counts=True, image_seed=60)
    print("cfg3 full length, second draw:", {k: {a: (f"{b:.2e}" if isinstance(b, float) else b) for a, b in v.items() if a != "trace"} for k, v in res.items()})
    for r in res.values():
        assert r["copies_identical"]
    _assert_cfg3_run(res, d)


def test_cfg3_two_distinct_images_share_the_batch_wide_cg_stop(dev):
    """The NON-degenerate batch: two different images (one of them much darker) through the same loop at 256 x 256 / 360 angles
    (tests/golden/cfg3_pair.npz).  The reference stops its CG when `torch.all(residual < tol)` holds over the BATCH
    (deepinv/optim/linear/conjugate_gradient.py:61) - the slower unit decides for both - and the device-side flag of
    dinv_cg_check reproduces exactly that: both reconstructions within 1e-4, per-prox A^T A counts as the reference's."""
    import deepinv_amd as dinv

    if not os.path.exists(os.path.join(G, "cfg3_pair.npz")):
        pytest.skip("tests/golden/cfg3_pair.npz not generated (tests/golden/make_golden_r6.py cfg3_pair)")
    d = load("cfg3_pair")
    W = int(d["width"])
    xs = torch.cat((torch.rand(1, 1, W, W, generator=gen(64)), 0.5 * torch.rand(1, 1, W, W, generator=gen(65)) ** 2))
    res = full_length_cfg3(dinv, dev, d, exact_cg_counts=True, images=xs)
    print("cfg3 pair:", {k: {a: (f"{b:.2e}" if isinstance(b, float) else b) for a, b in v.items() if a != "trace"} for k, v in res.items()})
    _assert_cfg3_run(res, d)


def full_length_cfg5(dinv, dev, d, B=16, precisions=None, runs=1):
    """BASELINE configs[4] at FULL length on the per-GPU shard (16 images 3x256x256): 100-step DiffPIR, unit 0 = the fixture's
    seeded image with the fixture's torch.randn_like draws replayed (the other units get different images and draws).  Returns
    {precision: (rel. error vs the reference's sample, vs the fp64 evaluation of the same sample path, trace errors)}.
    Shared by the GPU test below and by bench.py's cfg5 loop row (checker only)."""
    from oracle import drunet_cpu as OD

    st, stt = int(d["stride"]), int(d["stride_trace"])
    img = (3, 256, 256)
    # seeds of unit 0 (image, -, measurement noise, draws along the path): the fixture's own, cfg5_full.npz predates the field
    s_img, _, s_noise, s_draw = (int(v) for v in d["seeds"]) if "seeds" in d else (70, 72, 73, 74)
    x = torch.cat((torch.rand(1, *img, generator=gen(s_img)), torch.rand(B - 1, *img, generator=gen(700))))
    p = dinv.physics.Downsampling(img_size=img, filter="bicubic", factor=4, padding="circular", device=dev,
                                  noise_model=dinv.physics.GaussianNoise(0.05))
    y = p.A(x.to(dev))
    assert rel_err(y[:1], d["y"]) < TOL
    yn = torch.cat((d["y"] + 0.05 * torch.randn(1, 3, 64, 64, generator=gen(s_noise)),
                    y[1:].cpu() + 0.05 * torch.randn(B - 1, 3, 64, 64, generator=gen(703)))).to(dev)
    den = dinv.models.DRUNet(3, 3, pretrained=None).to(dev).eval()
    den.load_state_dict(OD.init_state_dict(3, 3, seed=int(d["drunet_seed"])))
    res = {}
    orig = torch.randn_like
    from deepinv_amd.models.drunet import CONV_PRECISIONS
    for prec in precisions or CONV_PRECISIONS:
        den.conv_precision = prec
        import time
        sampler = dinv.sampling.DiffPIR(den, dinv.optim.L2(), sigma=0.05, max_iter=int(d["steps"]), zeta=0.1, lambda_=7.0, device=dev)
        assert torch.equal(sampler.seq.cpu(), d["seq"])
        for _ in range(runs):       # (the last run is the one timed and compared: the first also packs weights and allocates)
            trace = []
            hook = den.register_forward_hook(lambda m, i, o: trace.append(o[:1].detach().reshape(-1)[::stt]))
            # the Gaussian draws of the whole run, made BEFORE the timed region (unit 0 replays the reference's generator stream,
            # the rest of the shard has its own): resident in HBM like every other input
            # (unit 0 on the CPU generator whose stream the fixture recorded; the other units' draws are compared with nothing and come
            # from the device's generator - 100 host draws of 15 images were 8 s of every run of this helper)
            g0, g1 = gen(s_draw), torch.Generator(device=dev).manual_seed(704)
            pool = [torch.cat((torch.randn(1, *img, generator=g0).to(dev), torch.randn(B - 1, *img, generator=g1, device=dev)))
                    for _ in range(int(d["steps"]) + 1)]
            pool.reverse()

            def draws(t, **kw):
                assert tuple(t.shape) == (B, *img)
                return pool.pop()

            torch.randn_like = draws
            try:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                out = sampler(yn, p)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
            finally:
                torch.randn_like = orig
                hook.remove()
        trace = [t.cpu() for t in trace]
        tr = torch.stack(trace)
        res[prec] = {"vs_reference": rel_err(sub(out[:1], st), d["out"]), "vs_fp64": rel_err(sub(out[:1], st), d["out_exact"]),
                     "trace_vs_fp64_max": max(rel_err(a, b) for a, b in zip(tr, d["den_outs_exact"])), "seconds": dt,
                     "finite": bool(torch.isfinite(out).all())}
    return res


def test_cfg5_diffpir_full_length_100_steps(dev):
    """configs[4] at its FULL length: 100 DiffPIR steps (deepinv/sampling/diffusion.py:423-513) on the per-GPU shard of 16 images,
    unit 0 against the REAL reference's sample (tests/golden/cfg5_full.npz, make_golden_r5.py) and against the fp64 evaluation
    of the same sample path, in both conv_precision settings; the denoiser output of every step is compared as well"""
    import deepinv_amd as dinv

    d = load("cfg5_full")
    res = full_length_cfg5(dinv, dev, d)
    print("cfg5 full length:", {k: {a: (f"{b:.2e}" if isinstance(b, float) else b) for a, b in v.items()} for k, v in res.items()})
    for prec, r in res.items():
        assert r["finite"]
        # the reference's own fp32 rounding along the path is `out_err_vs_exact` (2.5e-6 after 100 steps)
        assert r["vs_reference"] < max(TOL, 2.0 * float(d["out_err_vs_exact"])), (prec, r)
        assert r["vs_fp64"] < TOL, (prec, r)
        assert r["trace_vs_fp64_max"] < 10 * TOL, (prec, r)     # (early steps: x0 predictions of nearly pure noise, looser)


def test_cfg5_full_length_second_draw(dev):
    """A SECOND full-length draw of configs[4] against the real reference (tests/golden/cfg5_full_b.npz, make_golden_r6.py: cfg5_b -
    another image, DRUNet initialisation, measurement noise and Gaussian draws along the 100 steps).  Same assertions as the first."""
    import deepinv_amd as dinv

    if not os.path.exists(os.path.join(G, "cfg5_full_b.npz")):
        pytest.skip("tests/golden/cfg5_full_b.npz not generated (tests/golden/make_golden_r6.py cfg5_b: ~40 min of CPU)")
    d = load("cfg5_full_b")
    res = full_length_cfg5(dinv, dev, d)
    print("cfg5 full length, second draw:", {k: {a: (f"{b:.2e}" if isinstance(b, float) else b) for a, b in v.items()} for k, v in res.items()})
    for prec, r in res.items():
        assert r["finite"]
        assert r["vs_reference"] < max(TOL, 2.0 * float(d["out_err_vs_exact"])), (prec, r)
        assert r["vs_fp64"] < TOL, (prec, r)
        assert r["trace_vs_fp64_max"] < 10 * TOL, (prec, r)


@pytest.mark.parametrize("gain_tag", ["", "_gain"])
def test_cfg2_multicoil_pnp_pgd_320(dev, gain_tag):
    """The HEADLINE configuration (BASELINE configs[1]) against the REAL reference (tests/golden/cfg2_named.npz, written by
    make_golden_r4.py through deepinv.physics.MultiCoilMRI and deepinv.optim.PGD): 8-coil 320x320 MultiCoilMRI A / A_adjoint
    and the 50-iteration PnP-PGD reconstruction with DRUNet(2->2), batch 32 exactly as bench.py builds it (unit 0 is the
    fixture's slice), in BOTH arithmetic settings of the denoiser - once with the reference's weight initialisation
    (ResBlock gain 0.2) and once with O(1)-gain ResBlock convolutions, where the ResBlock kernels' rounding is not hidden
    behind the identity path"""
    import deepinv_amd as dinv
    from bench import make_problem
    from oracle import drunet_cpu as OD

    d = load("cfg2_named")
    st, iters = int(d["stride"]), int(d["iters"])
    B, H, W, coils = 32, 320, 320, 8
    physics, x, y, _, _ = make_problem(dinv, B, 0, H, W, coils, dev)
    if gain_tag == "":
        y0 = physics.A(x)
        assert rel_err(sub(y0[:1], st), d["y"]) < TOL
        assert bool(((y0 == 0) == (physics.mask[:, :, None].expand_as(y0) == 0)).all())       # bit-exact mask indexing
        assert rel_err(sub(physics.A_adjoint(y)[:1], st), d["yadj"]) < TOL
    gain = None if gain_tag == "" else float(d["res_gain"])
    den = dinv.models.DRUNet(2, 2, pretrained=None).to(dev).eval()
    den.load_state_dict(OD.init_state_dict(2, 2, seed=int(d["drunet_seed"]), res_gain=gain))
    model = dinv.optim.PGD(data_fidelity=dinv.optim.L2(), prior=dinv.optim.PnP(den), stepsize=1.0, g_param=0.05,
                           max_iter=iters, early_stop=False)
    from deepinv_amd.models.drunet import CONV_PRECISIONS
    for prec in CONV_PRECISIONS:
        den.conv_precision = prec
        with torch.no_grad():
            rec = model(y, physics)
        assert torch.isfinite(rec).all()
        err = rel_err(sub(rec[:1], st), d["rec" + gain_tag])
        assert err < TOL, (prec, gain_tag, err)
        if gain_tag == "" and os.path.exists(os.path.join(G, "cfg2_slices.npz")):
            # seven more slices of the batch through the real reference (make_golden_r5.py cfg2_slices): with slice 0 a quarter of
            # the batch, spread over its whole range
            ds = load("cfg2_slices")
            errs = {int(i): rel_err(sub(rec[int(i):int(i) + 1], int(ds["stride"])), ds["rec"][k]) for k, i in enumerate(ds["slices"])}
            assert len(errs) == 7 and max(errs.values()) < TOL, (prec, errs)


def test_cfg2_second_draw_other_maps_mask_and_initialisation(dev):
    """A second draw of the HEADLINE loop with everything re-drawn (tests/golden/cfg2_b.npz, make_golden_r6.py: cfg2_b - other coil
    maps, a 60-spoke mask, other images and noise, another DRUNet initialisation, g_param 0.08): four slices through the real
    reference's deepinv.optim.PGD, 50 iterations, both arithmetic settings."""
    import deepinv_amd as dinv
    from oracle import drunet_cpu as OD

    if not os.path.exists(os.path.join(G, "cfg2_b.npz")):
        pytest.skip("tests/golden/cfg2_b.npz not generated (tests/golden/make_golden_r6.py cfg2_b: ~3 min of CPU)")
    d = load("cfg2_b")
    st, iters, nsl, H, W, coils = int(d["stride"]), int(d["iters"]), int(d["slices"]), 320, 320, 8
    maps = torch.randn(1, coils, H, W, dtype=torch.complex64, generator=gen(int(d["maps_seed"])))
    maps = maps / maps.abs().pow(2).sum(dim=1, keepdim=True).sqrt()
    physics = dinv.physics.MultiCoilMRI(mask=dinv.utils.radial_mask(H, W, int(d["spokes"])), coil_maps=maps, img_size=(2, H, W), device=dev)
    ys = []
    for i in range(nsl):
        gi = gen(int(d["slice_seed0"]) + i)
        x = torch.rand(1, 2, H, W, generator=gi).to(dev)
        noise = torch.randn(1, 2, coils, H, W, generator=gi).to(dev)
        ys.append(physics.A(x) + 0.01 * noise * physics.mask[:, :, None])
    y = torch.cat(ys)
    assert rel_err(sub(y[:1], st), d["y0"]) < TOL
    den = dinv.models.DRUNet(2, 2, pretrained=None).to(dev).eval()
    den.load_state_dict(OD.init_state_dict(2, 2, seed=int(d["drunet_seed"])))
    model = dinv.optim.PGD(data_fidelity=dinv.optim.L2(), prior=dinv.optim.PnP(den), stepsize=1.0, g_param=float(d["g_param"]),
                           max_iter=iters, early_stop=False)
    from deepinv_amd.models.drunet import CONV_PRECISIONS
    for prec in CONV_PRECISIONS:
        den.conv_precision = prec
        with torch.no_grad():
            rec = model(y, physics)
        errs = [rel_err(sub(rec[k:k + 1], st), d["rec"][k]) for k in range(nsl)]
        print("cfg2 second draw", prec, ["%.2e" % e for e in errs])
        assert torch.isfinite(rec).all() and max(errs) < TOL, (prec, errs)
