"""Host logic of the iteration loop on CPU with a plain matrix LinearPhysics (no kernels involved):
the loop, parameter schedules, CG / least squares, unfolded trainable parameters.
Mirrors reference tests: optim/optimizers.py:179-218 doctest, test_optim.py:373-465 (optimality
condition), :1131-1180 (least-squares solvers)."""
import os

import pytest
import torch

import deepinv_amd as dinv


class MatPhysics(dinv.physics.LinearPhysics):
    def __init__(self, M):
        super().__init__()
        self.M = M

    def A(self, x, **kw):
        return x @ self.M.T

    def A_adjoint(self, y, **kw):
        return y @ self.M


def test_pgd_doctest_two_vector():
    """optimizers.py:179-218: min 1/2||Ax-y||^2 with A=diag(2,3), y=(2,3) -> x=(1,1)"""
    phys = MatPhysics(torch.tensor([[2.0, 0.0], [0.0, 3.0]]))
    y = torch.tensor([[2.0, 3.0]])
    for algo, kw in ((dinv.optim.PGD, dict(stepsize=0.1, max_iter=500)), (dinv.optim.HQS, dict(stepsize=10.0, max_iter=200))):
        x = algo(data_fidelity=dinv.optim.L2(), **kw)(y, phys)
        assert torch.allclose(x, torch.ones(1, 2), atol=1e-3), algo.__name__


def test_pgd_l1_optimality_condition():
    """first-order optimality of lasso solved by PGD with an explicit L1 prox (test_optim.py:373-465 spirit)"""
    torch.manual_seed(0)
    M = torch.randn(6, 4, dtype=torch.float64)
    phys = MatPhysics(M)
    y = torch.randn(1, 6, dtype=torch.float64)
    lam = 0.3

    class L1(dinv.optim.Prior):
        def __init__(self):
            super().__init__()
            self.explicit_prior = True

        def fn(self, x, *a, **k):
            return x.abs().sum(dim=-1)

        def prox(self, x, *a, gamma=1.0, **k):
            return torch.sign(x) * torch.clamp(x.abs() - gamma, min=0)

    step = 0.9 / float(torch.linalg.matrix_norm(M, 2) ** 2)
    model = dinv.optim.PGD(data_fidelity=dinv.optim.L2(), prior=L1(), lambda_reg=lam, stepsize=step, max_iter=5000,
                           early_stop=True, thres_conv=1e-13)
    x = model(y, phys)
    grad = phys.A_adjoint(phys.A(x) - y)
    # -grad in lam * subdifferential of |.|_1
    nz = x.abs() > 1e-9
    assert torch.allclose(-grad[nz], lam * torch.sign(x[nz]), atol=1e-6)
    assert torch.all(grad[~nz].abs() <= lam + 1e-6)


def test_params_schedules_and_metrics():
    phys = MatPhysics(torch.eye(3))
    y = torch.ones(2, 3)
    steps = [0.5, 0.25, 0.125]
    model = dinv.optim.PGD(data_fidelity=dinv.optim.L2(), stepsize=steps, max_iter=3)
    x, metrics = model(y, phys, x_gt=torch.ones(2, 3), compute_metrics=True)
    assert len(metrics["residual"]) == 2 and len(metrics["residual"][0]) == 3 and len(metrics["psnr"][0]) == 4
    with pytest.raises(ValueError):
        dinv.optim.PGD(data_fidelity=dinv.optim.L2(), stepsize=[0.1, 0.2], max_iter=5)
    # custom init (tensor / tuple / callable)
    x2 = dinv.optim.PGD(data_fidelity=dinv.optim.L2(), stepsize=0.0, max_iter=1, custom_init=lambda y, p: 3 * y)(y, phys)
    assert torch.allclose(x2, 3 * y)


def test_cg_least_squares_matches_dense_solve():
    torch.manual_seed(1)
    M = torch.randn(8, 5, dtype=torch.float64)
    phys = MatPhysics(M)
    y = torch.randn(3, 8, dtype=torch.float64)
    z = torch.randn(3, 5, dtype=torch.float64)
    gamma = 0.7
    x = dinv.optim.least_squares(phys.A, phys.A_adjoint, y, z=z, init=z, gamma=gamma, parallel_dim=[0], max_iter=200,
                                 tol=1e-12)
    H = M.T @ M + torch.eye(5, dtype=torch.float64) / gamma
    ref = torch.linalg.solve(H, (y @ M + z / gamma).T).T
    assert torch.allclose(x, ref, atol=1e-8)
    # pseudo-inverse branch (gamma=None, overcomplete): A^+ y
    xd = dinv.optim.least_squares(phys.A, phys.A_adjoint, y, parallel_dim=[0], max_iter=200, tol=1e-12)
    assert torch.allclose(xd, (torch.linalg.pinv(M) @ y.T).T, atol=1e-7)
    # prox_l2 / A_dagger of LinearPhysics route through it, with implicit backward
    phys.max_iter, phys.tol = 200, 1e-12
    zz = z.clone().requires_grad_(True)
    p = phys.prox_l2(zz, y, gamma)
    assert torch.allclose(p, ref, atol=1e-7)
    p.sum().backward()
    assert torch.allclose(zz.grad, torch.linalg.solve(H, torch.ones(5, 3, dtype=torch.float64)).T / gamma, atol=1e-6)


def test_unfolded_builder_trains():
    """deepinv/tests/test_unfolded.py:25-125: parameters are registered, receive gradients and change"""
    torch.manual_seed(2)
    M = torch.randn(6, 4) / 3
    phys = MatPhysics(M)
    x_true = torch.randn(5, 4)
    y = phys.A(x_true)

    class TinyDenoiser(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.lin = torch.nn.Linear(4, 4)

        def forward(self, x, sigma):
            return x + 0.1 * sigma * self.lin(x)

    model = dinv.unfolded.unfolded_builder("PGD", data_fidelity=dinv.optim.L2(), prior=dinv.optim.PnP(TinyDenoiser()),
                                           params_algo={"stepsize": 0.05, "g_param": 0.1, "lambda": 1.0}, max_iter=4,
                                           trainable_params=["stepsize", "g_param"])
    names = [n for n, _ in model.named_parameters()]
    assert "init_params_algo.stepsize.0" in names and "init_params_algo.g_param.0" in names
    assert any(n.startswith("prior.0.denoiser") for n in names)
    opt = torch.optim.SGD(model.parameters(), lr=1e-3)
    before = [p.detach().clone() for p in model.parameters()]
    loss0 = None
    for _ in range(5):
        opt.zero_grad()
        loss = (model(y, phys) - x_true).pow(2).mean()
        loss0 = loss.item() if loss0 is None else loss0
        loss.backward()
        opt.step()
    assert all(p.grad is not None for p in model.parameters())
    assert any(not torch.equal(a, b.detach()) for a, b in zip(before, model.parameters()))
    assert loss.item() < loss0


def test_dpir_schedule():
    s, step, n = dinv.optim.get_DPIR_params(0.05)
    assert n == 8 and abs(float(s[0]) - 49 / 255) < 1e-6 and abs(float(s[-1]) - 0.05) < 1e-6
    assert torch.allclose(step, (1 / 0.23) * (s / 0.05) ** 2)


def test_loop_options_match_reference_golden():
    """plain PGD and backtracking (optimizers.py:655-701) against vectors produced by the real
    reference (tests/golden/make_golden_optim.py): same lasso problem, matrix physics, float64, 12 iterations."""
    import os

    import numpy as np

    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "optim_loop_options.npz"))
    M, y, step = torch.from_numpy(gold["M"]), torch.from_numpy(gold["y"]), float(gold["step"])
    phys = MatPhysics(M)

    class L1(dinv.optim.Prior):
        def __init__(self):
            super().__init__()
            self.explicit_prior = True

        def fn(self, x, *a, **k):
            return x.abs().sum(dim=-1)

        def prox(self, x, *a, gamma=1.0, **k):
            return torch.sign(x) * torch.clamp(x.abs() - gamma, min=0)

    BT = dinv.optim.BacktrackingConfig
    cases = {
        "pgd_plain": (dict(), 1.0),
        "pgd_backtracking": (dict(backtracking=BT(gamma=0.1, eta=0.5, max_iter=20)), 8.0),
    }
    for name, (kw, scale) in cases.items():
        model = dinv.optim.PGD(data_fidelity=dinv.optim.L2(), prior=L1(), lambda_reg=0.2, stepsize=step * scale,
                               max_iter=12, early_stop=False, **kw)
        with torch.no_grad():
            x = model(y, phys)
        ref = torch.from_numpy(gold[name])
        assert torch.allclose(x, ref, rtol=1e-9, atol=1e-11), (name, float((x - ref).abs().max()))


def test_two_reconstructions_do_not_share_the_adjoint():
    """Regression for the round-1 stale A^T y bug: A^T y is reused inside ONE call only.  A fresh measurement that
    the allocator places at the freed address of the previous one (same shape, same version counter) must be
    reconstructed from its own A^T y (reference: recomputed every iteration, data_fidelity.py:335-338)."""
    g = torch.Generator().manual_seed(3)
    M = torch.randn(6, 4, generator=g)
    phys = MatPhysics(M)
    step = 0.9 / float(torch.linalg.matrix_norm(M, 2) ** 2)
    model = dinv.optim.PGD(data_fidelity=dinv.optim.L2(), stepsize=step, max_iter=15)
    import numpy as np

    arr = np.empty((3, 6), np.float32)                # one block of memory, rewritten behind torch's back
    ptrs, ys, xs = [], [], []
    for k in range(20):
        arr[...] = torch.randn(3, 6, generator=g).numpy()
        y = torch.from_numpy(arr)                     # a NEW tensor: same address, same shape, version counter 0
        assert y._version == 0
        ptrs.append(y.data_ptr())
        xs.append(model(y, phys))
        assert model.fixed_point.call_ctx is None and model.fixed_point.iterator.f_step.call_ctx is None
        ys.append(y.clone())
        del y
    reused = sum(int(a == b) for a, b in zip(ptrs[1:], ptrs[:-1]))
    for k, (y, x) in enumerate(zip(ys, xs)):
        ref = y @ M                                    # the reference's loop, written out
        for _ in range(15):
            ref = ref - step * (ref @ M.T @ M - y @ M)
        assert torch.allclose(x, ref, rtol=1e-5, atol=1e-6), k
    assert reused == 19   # the scenario really occurred


def test_adjoint_of_y_evaluated_once_per_call():
    """one A^T y per reconstruction in no-grad mode (reference: 2 + max_iter); every iteration still applies A^T A"""
    calls = {"adj": 0}

    class Counting(MatPhysics):
        def A_adjoint(self, y, **kw):
            calls["adj"] += 1
            return super().A_adjoint(y, **kw)

    phys = Counting(torch.randn(5, 3))
    y = torch.randn(2, 5)
    dinv.optim.PGD(data_fidelity=dinv.optim.L2(), stepsize=0.01, max_iter=7)(y, phys)
    assert calls["adj"] == 1 + 7          # one for A^T y, one inside A_adjoint_A per iteration
    calls["adj"] = 0
    with torch.inference_mode():          # tensors without version counters must work too
        dinv.optim.PGD(data_fidelity=dinv.optim.L2(), stepsize=0.01, max_iter=7)(torch.randn(2, 5), phys)
    assert calls["adj"] == 1 + 7


def _gold(name):
    import os

    import numpy as np

    d = np.load(os.path.join(os.path.dirname(__file__), "golden", name + ".npz"))
    return {k: torch.from_numpy(d[k]) for k in d.files}


def test_unfolded_builder_host_logic_matches_reference_golden():
    """The host side of unfolded_builder("PGD") (parameter wrapping, graph through the loop) against loss + gradients
    produced by the REAL reference (tests/golden/make_golden_r2.py); physics = the CPU oracle operators."""
    from oracle import physics_cpu as O

    d = _gold("unfolded_pgd")
    maps = torch.view_as_complex(d["maps"].contiguous())

    class PhysCPU(dinv.physics.LinearPhysics):
        def A(self, v, **k):
            return O.multicoil_A(v, maps, d["mask"], True)

        def A_adjoint(self, v, **k):
            return O.multicoil_AT(v, maps, d["mask"], True)

    class Den(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.c = torch.nn.Conv3d(2, 2, 3, padding=1)
            with torch.no_grad():
                self.c.weight.copy_(d["wden"])
                self.c.bias.copy_(d["bden"])

        def forward(self, u, s):
            return u - s * self.c(u)

    model = dinv.unfolded.unfolded_builder("PGD", data_fidelity=dinv.optim.L2(), prior=dinv.optim.PnP(Den()),
                                           params_algo={"stepsize": 0.8, "g_param": 0.05, "lambda": 1.0}, max_iter=3,
                                           trainable_params=["stepsize", "g_param"])
    rec = model(d["y"], PhysCPU())
    loss = (rec - d["x"]).pow(2).mean()
    loss.backward()
    assert torch.allclose(rec, d["rec"], rtol=1e-5, atol=1e-6)
    assert abs(float(loss) - float(d["loss"])) / float(d["loss"]) < 1e-5
    for n, p in model.named_parameters():
        ref = d["grad_" + n.replace(".", "_")]
        assert float((p.grad - ref).norm() / ref.norm()) < 1e-4, n


def test_diffpir_schedule_matches_reference_golden():
    """DiffPIR.get_alpha_beta / get_noise_schedule (diffusion.py:323-375): host-side schedule of the product against the
    REAL reference's, three settings (this is pure host logic: no kernels involved)."""
    for name, kw in (("diffpir", dict(sigma=0.05, max_iter=6, lambda_=7.0)), ("diffpir_schedule_a", None),
                     ("diffpir_schedule_b", None)):
        d = _gold(name)
        if kw is None:
            kw = dict(sigma=float(d["sigma"]), max_iter=int(d["max_iter"]), lambda_=float(d["lambda_"]))
        s = dinv.sampling.DiffPIR(None, None, zeta=0.1, device="cpu", **kw)
        rhos, sigmas, seq = s.get_noise_schedule(sigma=torch.tensor(kw["sigma"]))   # as forward() does (:441-443)
        assert torch.equal(seq, d["seq"])
        assert torch.allclose(sigmas, d["sigmas"], rtol=1e-6) and torch.allclose(rhos, d["rhos"], rtol=1e-6)
        if "reduced" in d:
            assert torch.allclose(s.reduced_alpha_cumprod, d["reduced"], rtol=1e-7)


@pytest.mark.parametrize("kw", [dict(sigma=0.05, max_iter=6, lambda_=7.0), dict(sigma=0.1, max_iter=30, lambda_=3.0),
                                dict(sigma=0.02, max_iter=100, lambda_=7.0)])
def test_diffpir_host_schedule_equals_the_per_step_lookups(kw):
    """DiffPIR._host_schedule (the per-step scalars, evaluated once per call) against the reference's per-step tensor look-ups
    (diffusion.py:452-507) on the same schedule tensors: identical time steps, scalars equal to fp32 rounding, and the fused
    update x <- cx x + cx0 (2 x0_p - 1) + cn n equal to the reference's three-line form"""
    zeta = 0.1
    s = dinv.sampling.DiffPIR(None, None, zeta=zeta, device="cpu", **kw)
    steps = s._host_schedule()
    sr, _ = s.get_alpha_prod()
    n = len(s.seq)
    assert len(steps) == n and steps[-1]["last"]
    g = torch.Generator().manual_seed(0)
    for i, st in enumerate(steps):
        cs = s.sigmas[s.seq[i]]
        t_i = s.find_nearest(s.reduced_alpha_cumprod, cs)
        at = 1 / sr[t_i] ** 2
        assert st["sigma_den"] == pytest.approx(float(cs / 2), rel=1e-7)
        assert st["pre_scale"] == pytest.approx(float(1 / (2 * at.sqrt())), rel=1e-6)
        assert st["last"] == bool(s.seq[i] == s.seq[-1])
        if i == 0:
            assert st["init_noise"] == pytest.approx(float((cs ** 2 - 4.0 * kw["sigma"] ** 2).sqrt()), rel=1e-6)
            assert st["init_div"] == pytest.approx(float(sr[-1]), rel=1e-7)
        if st["last"]:
            continue
        assert st["gamma"] == pytest.approx(float(1.0 / (2 * s.rhos[t_i])), rel=1e-6)
        t_im1 = s.find_nearest(s.reduced_alpha_cumprod, s.sigmas[s.seq[i + 1]])
        x, x0p, nz = (torch.randn(5, generator=g).double() for _ in range(3))
        x0 = x0p * 2 - 1
        eps = (x - s.sqrt_alphas_cumprod[t_i].double() * x0) / s.sqrt_1m_alphas_cumprod[t_i].double()
        ref = (s.sqrt_alphas_cumprod[t_im1].double() * x0 + s.sqrt_1m_alphas_cumprod[t_im1].double() * (1 - zeta) ** 0.5 * eps
               + s.sqrt_1m_alphas_cumprod[t_im1].double() * zeta ** 0.5 * nz)
        fused = st["cx"] * x + 2.0 * st["cx0"] * x0p + st["cn"] * nz - st["cx0"]
        assert torch.allclose(fused, ref, rtol=1e-9, atol=1e-9)


@pytest.mark.parametrize("solver", ["CG", "BiCGStab", "lsqr", "minres"])
def test_least_squares_solvers_match_reference_golden(solver):
    """least_squares(solver=...) (least_squares.py:15-197, lsqr.py, bicgstab.py, minres.py) against outputs of the REAL
    reference on tall / wide / square batched matrix operators, with no, scalar and per-sample gamma."""
    d = _gold("ls_solvers")
    for tag in ("tall", "wide", "square"):
        M, y, z = d[f"{tag}_M"], d[f"{tag}_y"], d[f"{tag}_z"]
        A = lambda x: torch.einsum("bij,bj->bi", M, x)
        AT = lambda v: torch.einsum("bij,bi->bj", M, v)
        for gname, gamma in (("none", None), ("scalar", 0.7), ("batched", torch.tensor([0.3, 1.0, 2.5], dtype=torch.float64))):
            x = dinv.optim.least_squares(A, AT, y, z=z if gamma is not None else 0.0, init=z if gamma is not None else None,
                                         gamma=gamma, solver=solver, max_iter=200, tol=1e-10)
            ref = d[f"{tag}_{solver}_{gname}"]
            assert torch.allclose(x, ref, rtol=1e-7, atol=1e-9), (tag, solver, gname, float((x - ref).abs().max()))


def test_fan_beam_host_tables_reproduce_the_reference_grid():
    """deepinv_amd.hip.radon.fan_tables (pure torch, host side) + the kernels' coordinate formula
    R(theta) (xm_i, yd_d * sc_i) == fan_beam_grid of the reference (functional/radon.py:16-52, restated and golden-pinned
    in oracle.physics_cpu.fan_grid), for the default and a custom geometry"""
    import torch
    from deepinv_amd.hip.radon import fan_tables
    from oracle import physics_cpu as O

    for W, G, fan in ((16, 23, None), (40, 57, {"pixel_spacing": 0.05, "source_radius": 3.0, "detector_radius": 5.0,
                                                "n_detector_pixels": 47, "detector_spacing": 0.11})):
        fp, xm, sc, yd = fan_tables(G, W, fan)
        assert fp == O.fan_parameters_filled(W, fan)
        for deg in (0.0, 33.0, 91.0, 270.5):
            th = O._deg2rad(torch.tensor(deg))
            ref = O.fan_grid(th, G, fp)[0]                       # [G (march), n_det, 2]
            c, s = th.cos(), th.sin()
            px = xm[:, None].expand(-1, yd.numel())
            py = yd[None, :] * sc[:, None]
            mine = torch.stack((c * px + s * py, -s * px + c * py), dim=-1)
            assert float((mine - ref).abs().max()) < 2e-6 * max(1.0, float(ref.abs().max()))


def test_overlap_tiling_edge_cases():
    from deepinv_amd.distributed import OverlapTiling

    t = OverlapTiling((1, 1, 10, 50), patch_size=64, overlap=4)          # patch larger than the signal: one window
    assert len(t) == 1 and t.patch == [10, 50]
    t = OverlapTiling((2, 3, 40), patch_size=16, overlap=3, tiling_dims=-1)       # 1-D tiling
    assert len(t) == 3 and [w[0][:1] for w in t.windows] == [(0,), (16,), (24,)]   # last window shifted inwards
    import pytest
    with pytest.raises(ValueError):
        OverlapTiling((1, 1, 32, 32), patch_size=(16, 16, 16), overlap=2, tiling_dims=(-2, -1))


def test_trainer_loop_host():
    """deepinv_amd.Trainer's call sequence on CPU tensors with a toy linear physics (no kernels involved): two epochs of SGD
    on a least-squares 'network' reproduce the hand-written update, the scheduler steps once per epoch, checkpoints load"""
    import tempfile

    import deepinv_amd as dinv

    torch.manual_seed(0)
    M = torch.randn(6, 4)

    class P(dinv.physics.LinearPhysics):
        def A(self, x, **k):
            return x @ M.T

        def A_adjoint(self, y, **k):
            return y @ M

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.w = torch.nn.Parameter(torch.tensor(0.05))

        def forward(self, y, physics, **k):
            return self.w * physics.A_adjoint(y)

    x = torch.randn(8, 4)
    y = x @ M.T
    data = [(x[i], y[i]) for i in range(8)]
    loader = torch.utils.data.DataLoader(data, batch_size=4, shuffle=False)
    net = Net()
    opt = torch.optim.SGD(net.parameters(), lr=0.01)
    sched = torch.optim.lr_scheduler.StepLR(opt, step_size=1, gamma=0.5)
    with tempfile.TemporaryDirectory() as tmp:
        tr = dinv.Trainer(model=net, physics=P(), optimizer=opt, train_dataloader=loader, epochs=2, device="cpu", verbose=False,
                          scheduler=sched, save_path=tmp, eval_dataloader=loader)
        tr.train()
        w, lr = torch.tensor(0.05), 0.01
        for epoch in range(2):
            for i in range(2):
                xb, yb = x[4 * i:4 * i + 4], y[4 * i:4 * i + 4]
                aty = yb @ M
                grad = (2 * (w * aty - xb) * aty).mean()
                w = w - lr * grad
            lr *= 0.5
        assert abs(float(net.w) - float(w)) < 1e-6
        assert len(tr.train_loss_history) == 2 and len(tr.eval_metric_history) == 2
        net2 = Net()
        tr2 = dinv.Trainer(model=net2, physics=P(), optimizer=torch.optim.SGD(net2.parameters(), lr=0.01), train_dataloader=loader,
                           epochs=2, device="cpu", verbose=False, ckpt_pretrained=os.path.join(tmp, "ckp_1.pth.tar"))
        tr2.setup_train()
        assert abs(float(net2.w) - float(net.w)) < 1e-7 and tr2.epoch_start == 2
        # the histories travel in the reference's checkpoint shape, {name: [value per epoch]} (trainer.py:1192-1198), and come back
        # as this trainer's flat lists - also from a checkpoint the reference wrote ({loss class: [..]} per loss, several metrics)
        ck = torch.load(os.path.join(tmp, "ckp_1.pth.tar"))
        assert isinstance(ck["loss"], dict) and isinstance(ck["eval_metrics"], dict)
        assert tr2.train_loss_history == pytest.approx(tr.train_loss_history) and tr2.eval_metric_history == pytest.approx(tr.eval_metric_history)
        ck["loss"] = {"SupLoss": [1.0, 0.5], "OtherLoss": [0.25, 0.125]}
        ck["eval_metrics"] = {"PSNR": [20.0, 21.0], "SSIM": [0.5, 0.6]}
        ck["train_metrics"], ck["eval_loss"] = {"PSNR": [19.0, 20.0]}, {"SupLoss": []}
        torch.save(ck, os.path.join(tmp, "ref_style.pth.tar"))
        tr3 = dinv.Trainer(model=Net(), physics=P(), optimizer=torch.optim.SGD(Net().parameters(), lr=0.01), train_dataloader=loader,
                           epochs=2, device="cpu", verbose=False, ckpt_pretrained=os.path.join(tmp, "ref_style.pth.tar"))
        tr3.setup_train()
        assert tr3.train_loss_history == [1.25, 0.625] and tr3.eval_metric_history == [20.0, 21.0]
    with pytest.raises(ValueError, match="tuple"):
        bad = dinv.Trainer(model=net, physics=P(), optimizer=opt, train_dataloader=torch.utils.data.DataLoader(list(x), batch_size=4),
                           epochs=1, device="cpu", verbose=False)
        bad.train()


def test_mri_crop_rescale_matches_the_defining_sums():
    """MRIMixin.crop(rescale=True) (deepinv/utils/mixins.py:208-246: torchvision Resize of the last two dims, odd heights adjusted
    by one pixel) against an fp64 evaluation of antialiased bilinear resampling (oracle/naive.py); crop and rescale exclude each
    other.  (No reference fixture: torchvision is not installed where the reference can be imported.)"""
    import numpy as np
    import pytest

    from deepinv_amd.physics.mri import MRIMixin
    from oracle.naive import resize_bilinear_antialias

    m = MRIMixin()
    x = torch.randn(2, 3, 40, 52, generator=torch.Generator().manual_seed(3))
    for shape, out_shape in (((20, 26), (20, 26)), ((64, 70), (64, 70)), ((25, 30), (25, 30)), ((40, 52), (40, 52))):
        m.img_size = (2, *shape)
        out = m.crop(x, crop=False, rescale=True)
        assert tuple(out.shape) == (2, 3, *out_shape)
        hh = shape[0] + (shape[0] % 2)
        ref = resize_bilinear_antialias(x.numpy(), hh, shape[1])[..., :shape[0], :]
        assert np.abs(out.numpy() - ref).max() < 5e-5      # (ATen evaluates in fp32)
    with pytest.raises(ValueError):
        m.crop(x, crop=True, rescale=True)
    assert m.crop(x, crop=False) is x
