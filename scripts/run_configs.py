"""BASELINE.json configs[2..4] at their NAMED shapes on one MI355X (per-GPU shard of each config):
parity spot-check against the CPU oracle, operator times (algorithmic GB/s, samples/s), end-to-end loop time.

    python scripts/run_configs.py [cfg3] [cfg4] [cfg5] [--no-parity] > gpurun_out/configs.jsonl

One JSON object per line.  The oracle (oracle/*.py) is used here as the CHECKER only.
  cfg3  Tomography 512x512, 720 angles, FBP init + 30-iteration PnP-HQS (DRUNet 1->1), 8 images (= 64 / 8 GPUs)
  cfg4  3-D MultiCoilMRI 12 coils 16x256x256, unfolded PGD 10 iterations fwd+bwd, 2 volumes (= 8 / 4 GPUs)
  cfg5  Downsampling x4 (bicubic, circular) on 3x256x256, DiffPIR 100 steps (DRUNet 3->3), 16 images (= 128 / 8 GPUs)
"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deepinv_amd as dinv  # noqa: E402

HBM_PEAK = 8.0e12
DEV = torch.device("cuda:0")
PARITY = "--no-parity" not in sys.argv


def emit(**kw):
    print(json.dumps(kw), flush=True)


def timeit(fn, iters=10, warmup=2):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).norm() / b.norm())


def cfg3():
    from oracle import physics_cpu as O
    B, W, nang = 8, 512, 720
    t0 = time.perf_counter()
    phys = dinv.physics.Tomography(angles=nang, img_width=W, circle=False, normalize=True, device=DEV)
    torch.cuda.synchronize()
    t_init = time.perf_counter() - t0
    g = torch.Generator().manual_seed(0)
    x = torch.rand(B, 1, W, W, generator=g).to(DEV)
    y = phys.A(x)
    G = y.shape[2]
    alg = B * (W * W + G * nang) * 4
    if PARITY:
        ang = phys.angles.cpu()
        nrm = phys.operator_norm.cpu()
        t0 = time.perf_counter()
        y_ref = O.radon_forward(x[:1].cpu(), ang) / nrm          # sequential CPU oracle, one image
        t_cpu = time.perf_counter() - t0
        v = torch.randn(1, 1, G, nang, generator=g)
        xa = phys.A_adjoint(v.to(DEV))
        # adjoint against the ORACLE's forward: <A_oracle x, v> == <x, A^T_hip v>
        lhs = float((y_ref.double() * v.double()).sum())
        rhs = float((x[:1].cpu().double() * xa.cpu().double()).sum())
        ramp_err = rel(phys.filter(y[:1]), O.ramp_filter(y[:1].cpu()))
        emit(cfg=3, check="parity@512x512x720", A_rel_err=rel(y[:1], y_ref), cross_dot_rel=abs(lhs - rhs) / abs(lhs),
             ramp_rel_err=ramp_err, oracle_A_s_per_img=t_cpu, operator_norm=float(nrm), init_s=t_init)
    ops = (("A", lambda: phys.A(x), 3), ("A_adjoint", lambda: phys.A_adjoint(y), 3), ("ramp", lambda: phys.filter(y), 5),
           ("fbp", lambda: phys.A_dagger(y, fbp=True), 3))
    for name, fn, it in ops:
        t = timeit(fn, iters=it, warmup=1)
        emit(cfg=3, op=f"Tomography.{name}", B=B, W=W, angles=nang, ms=t * 1e3, ms_per_img=t * 1e3 / B, alg_MB=alg / 1e6,
             GBps=alg / t / 1e9, frac_hbm_peak=alg / t / HBM_PEAK, Gsamples_per_s=B * G * G * nang / t / 1e9)
    torch.manual_seed(0)
    den = dinv.models.DRUNet(1, 1, pretrained=None).to(DEV).eval()
    sig, steps, _ = dinv.optim.get_DPIR_params(0.02)
    import numpy as np
    n_it = 30   # DPIR schedule stretched to 30 iterations
    s30 = np.logspace(np.log10(49 / 255.0), np.log10(0.02), n_it).astype("float32")
    st30 = ((s30 / 0.02) ** 2 / 0.23).astype("float32")
    model = dinv.optim.HQS(data_fidelity=dinv.optim.L2(), prior=dinv.optim.PnP(den), stepsize=list(map(float, st30)),
                           g_param=list(map(float, s30)), max_iter=n_it, early_stop=False,
                           custom_init=lambda yy, p: p.A_dagger(yy, fbp=True))
    with torch.no_grad():
        model(y, phys)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rec = model(y, phys)
        torch.cuda.synchronize()
        t = time.perf_counter() - t0
    emit(cfg=3, loop="FBP + PnP-HQS 30 it (CG prox) + DRUNet(1->1)", B=B, s=t, images_per_s=B / t,
         finite=bool(torch.isfinite(rec).all()))


def cfg4():
    from oracle import physics_cpu as O
    B, coils, vol = 2, 12, (16, 256, 256)
    g = torch.Generator().manual_seed(0)
    x = torch.rand(B, 2, *vol, generator=g).to(DEV)
    maps = (torch.randn(1, coils, *vol, dtype=torch.complex64, generator=g) / coils ** 0.5).to(DEV)
    mask = torch.zeros(*vol)
    mask[..., ::4] = 1
    mask[..., 128 - 10:128 + 10] = 1
    mask = mask.to(DEV)
    phys = dinv.physics.MultiCoilMRI(mask=mask, coil_maps=maps, img_size=(2, *vol), three_d=True, device=DEV)
    y = phys.A(x)
    nvol = vol[0] * vol[1] * vol[2]
    alg = B * 2 * nvol * 4 + B * 2 * coils * nvol * 4 + coils * nvol * 8 + 2 * nvol * 4
    if PARITY:
        y_ref = O.multicoil_A(x.cpu(), maps.cpu(), mask.cpu(), True)
        xa_ref = O.multicoil_AT(y_ref, maps.cpu(), mask.cpu(), True)
        emit(cfg=4, check="parity@12x16x256x256,B=2", A_rel_err=rel(y, y_ref), AT_rel_err=rel(phys.A_adjoint(y), xa_ref),
             zero_pattern_exact=bool(((y == 0) == (phys.mask[:, :, None].expand_as(y) == 0)).all()))
    for name, fn in (("A", lambda: phys.A(x)), ("A_adjoint", lambda: phys.A_adjoint(y))):
        t = timeit(fn)
        emit(cfg=4, op=f"MultiCoilMRI3D.{name}", B=B, coils=coils, vol=vol, ms=t * 1e3, alg_MB=alg / 1e6, GBps=alg / t / 1e9,
             frac_hbm_peak=alg / t / HBM_PEAK)
    torch.manual_seed(0)
    den = dinv.models.DRUNet(2, 2, nc=(16, 32, 64, 128), nb=1, pretrained=None, dim=3).to(DEV)   # "3-D DRUNet small"
    model = dinv.unfolded.unfolded_builder("PGD", data_fidelity=dinv.optim.L2(), prior=dinv.optim.PnP(den),
                                           params_algo={"stepsize": 1.0, "g_param": 0.05, "lambda": 1.0}, max_iter=10,
                                           trainable_params=["stepsize", "g_param"], device=DEV).to(DEV)

    def step():
        for p in model.parameters():
            p.grad = None
        loss = (model(y, phys) - x).pow(2).mean()
        loss.backward()
        return loss

    step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    loss = step()
    torch.cuda.synchronize()
    t = time.perf_counter() - t0
    emit(cfg=4, loop="unfolded PGD 10 it, forward + backward, DRUNet3D(nc=16..128, nb=1)", B=B, s=t, volumes_per_s=B / t,
         loss=float(loss), grads_finite=all(bool(torch.isfinite(p.grad).all()) for p in model.parameters() if p.grad is not None))


def cfg5():
    from oracle import physics_cpu as O
    B, img, f = 16, (3, 256, 256), 4
    g = torch.Generator().manual_seed(0)
    x = torch.rand(B, *img, generator=g).to(DEV)
    phys = dinv.physics.Downsampling(img_size=img, filter="bicubic", factor=f, padding="circular", device=DEV,
                                     noise_model=dinv.physics.GaussianNoise(0.05))
    y = phys.A(x)
    algA = B * 3 * (256 * 256 + 64 * 64) * 4
    if PARITY:
        k = phys.filter.cpu()
        y_ref = O.downsampling_A(x.cpu(), k, f)
        z = torch.rand(B, *img, generator=g)
        p_ref = O.downsampling_prox_l2(z, y_ref, 0.7, k, f, img)
        emit(cfg=5, check="parity@16x3x256x256", A_rel_err=rel(y, y_ref),
             AT_rel_err=rel(phys.A_adjoint(y), O.downsampling_AT(y_ref, k, f, img)),
             prox_rel_err=rel(phys.prox_l2(z.to(DEV), y, 0.7), p_ref))
    z = torch.rand(B, *img, device=DEV)
    for name, fn, alg in (("A", lambda: phys.A(x), algA), ("A_adjoint", lambda: phys.A_adjoint(y), algA),
                          ("prox_l2", lambda: phys.prox_l2(z, y, 0.7), B * 3 * 256 * 256 * 4 * 2 + B * 3 * 64 * 64 * 4)):
        t = timeit(fn)
        emit(cfg=5, op=f"Downsampling.{name}", B=B, img=img, ms=t * 1e3, alg_MB=alg / 1e6, GBps=alg / t / 1e9,
             frac_hbm_peak=alg / t / HBM_PEAK)
    torch.manual_seed(0)
    den = dinv.models.DRUNet(3, 3, pretrained=None).to(DEV).eval()
    sampler = dinv.sampling.DiffPIR(den, dinv.optim.L2(), sigma=0.05, max_iter=100, zeta=0.1, lambda_=7.0, device=DEV)
    yn = phys(x)
    sampler(yn, phys, seed=0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = sampler(yn, phys, seed=0)
    torch.cuda.synchronize()
    t = time.perf_counter() - t0
    emit(cfg=5, loop="DiffPIR 100 steps + DRUNet(3->3)", B=B, s=t, images_per_s=B / t, finite=bool(torch.isfinite(out).all()))


if __name__ == "__main__":
    which = [a for a in sys.argv[1:] if not a.startswith("-")] or ["cfg3", "cfg4", "cfg5"]
    for name in which:
        try:
            {"cfg3": cfg3, "cfg4": cfg4, "cfg5": cfg5}[name]()
        except Exception as e:   # keep going: one broken config must not hide the others
            import traceback
            traceback.print_exc()
            emit(cfg=name, error=repr(e))
