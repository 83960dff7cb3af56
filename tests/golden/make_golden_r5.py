#!/usr/bin/env python
"""Round-5 golden vectors from the REAL reference (deepinv v0.4.1 at /root/reference through oracle/ref_shim.py): the loops of
BASELINE.json configs[2] and configs[4] at their FULL length (VERDICT r4 "next" #1; round 3's fixtures stop after 3 / 5 steps).

* `cfg3_full.npz`  one 512x512 image, `Tomography(angles=720, img_width=512, circle=False, normalize=True)` with the
                   reference's DEFAULT solver settings (`LinearPhysics(max_iter=50, tol=1e-4)`, CG), FBP-initialised
                   `deepinv.optim.HQS` (deepinv/optim/optimizers.py:1459-1593) for **30 iterations** with the DPIR-style
                   schedules of bench.py (`s30`, `st30`: deepinv/optim/dpir.py:22-35 stretched to 30), DRUNet(1->1):
                   the reconstruction, the denoiser output of every iteration (strided), and the number of `A_adjoint_A`
                   applications of every prox (CG iterations + 1).
* `cfg5_full.npz`  one 3x256x256 image, `Downsampling(x4, bicubic, circular)` + `GaussianNoise(0.05)`,
                   `deepinv.sampling.DiffPIR` (deepinv/sampling/diffusion.py:423-513) for **100 steps**, zeta 0.1, lambda 7,
                   DRUNet(3->3), every `torch.randn_like` draw regenerated from a seed: the sample, the denoiser output of
                   every step (strided), and the same sample path evaluated in fp64 through the oracle restatement.

* `cfg2_slices.npz` seven more slices (4, 9, 13, 18, 22, 27, 31) of the headline batch (configs[1]: 8-coil 320x320 MultiCoilMRI, 50-iteration
                   PnP-PGD, DRUNet(2->2)) through `deepinv.optim.PGD`: with slice 0 of cfg2_named.npz a quarter of the bench batch is
                   pinned to the reference, spread over its whole range.

Inputs are regenerated from seeds by the tests; large outputs are stored as strided subsamples flat[::stride].

    python tests/golden/make_golden_r5.py [cfg5] [cfg3]        # cfg5: ~10 min, cfg3: ~1-2 h on 8 cores
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.ref_shim import import_reference  # noqa: E402
from oracle import drunet_cpu as OD  # noqa: E402

dinv = import_reference()
OUT = os.path.dirname(os.path.abspath(__file__))
STRIDE = 7
STRIDE_TRACE = 1009          # per-iteration traces: ~260 / ~195 values each


def sub(t, stride=STRIDE):
    return t.detach().reshape(-1)[::stride].clone()


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().numpy()
        out[k] = np.asarray(v)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(name, {k: v.shape for k, v in out.items()}, flush=True)


def g(seed):
    return torch.Generator().manual_seed(seed)


def hqs_schedule(n=30):
    """bench.py: other_config_ops (cfg3): DPIR's log-spaced sigma schedule 49/255 -> 0.02 and stepsize (sigma/0.02)^2/0.23"""
    s = np.logspace(np.log10(49 / 255.0), np.log10(0.02), n).astype("float32")
    st = ((s / 0.02) ** 2 / 0.23).astype("float32")
    return s, st


class Trace(torch.nn.Module):
    """denoiser wrapper that keeps a strided copy of every output"""

    def __init__(self, den):
        super().__init__()
        self.den, self.outs = den, []

    def forward(self, x, sigma, *a, **k):
        o = self.den(x, sigma, *a, **k)
        self.outs.append(sub(o, STRIDE_TRACE))
        print("  denoiser call", len(self.outs), time.strftime("%H:%M:%S"), flush=True)
        return o


def cfg3():
    W, nang, iters = 512, 720, 30
    t0 = time.time()
    p = dinv.physics.Tomography(angles=nang, img_width=W, circle=False, normalize=True, device="cpu")
    print("cfg3 init", time.time() - t0, "max_iter", p.max_iter, "tol", p.tol, "solver", p.solver, flush=True)
    x = torch.rand(1, 1, W, W, generator=g(50))
    y = p.A(x)
    den = dinv.models.DRUNet(in_channels=1, out_channels=1, pretrained=None)
    den.load_state_dict(OD.init_state_dict(1, 1, seed=52))
    den.eval()
    tr = Trace(den)
    sigs, steps = hqs_schedule(iters)
    n_ata, cnt = [], [0]
    ata = p.A_adjoint_A

    def counting_ata(v, **kw):
        cnt[0] += 1
        return ata(v, **kw)

    p.A_adjoint_A = counting_ata
    prox = p.prox_l2

    def counting_prox(*a, **kw):
        cnt[0] = 0
        o = prox(*a, **kw)
        n_ata.append(cnt[0])
        print("  prox: A_adjoint_A applications", cnt[0], time.strftime("%H:%M:%S"), flush=True)
        return o

    p.prox_l2 = counting_prox
    model = dinv.optim.HQS(data_fidelity=dinv.optim.L2(), prior=dinv.optim.PnP(tr), stepsize=list(map(float, steps)),
                           g_param=list(map(float, sigs)), max_iter=iters, early_stop=False,
                           custom_init=lambda yy, pp: pp.A_dagger(yy, fbp=True))
    t0 = time.time()
    with torch.no_grad():
        rec = model(y, p)
    print("cfg3 loop", time.time() - t0, flush=True)
    save("cfg3_full", operator_norm=p.operator_norm, rec=sub(rec), den_outs=torch.stack(tr.outs), n_ata=np.int32(n_ata),
         steps=steps, sigs=sigs, stride=STRIDE, stride_trace=STRIDE_TRACE, drunet_seed=52, iters=iters,
         cg_max_iter=p.max_iter, cg_tol=p.tol)


def cfg5():
    img, f, steps = (3, 256, 256), 4, 100
    x = torch.rand(1, *img, generator=g(70))
    p = dinv.physics.Downsampling(img_size=img, filter="bicubic", factor=f, padding="circular",
                                  noise_model=dinv.physics.GaussianNoise(0.05))
    y = p.A(x)
    sd = OD.init_state_dict(3, 3, seed=72)
    den = dinv.models.DRUNet(in_channels=3, out_channels=3, pretrained=None)
    den.load_state_dict(sd)
    den.eval()
    tr = Trace(den)
    yn = y + 0.05 * torch.randn(1, 3, 64, 64, generator=g(73))
    gen = g(74)
    _orig = torch.randn_like
    torch.randn_like = lambda t, **kw: torch.randn(t.shape, generator=gen)
    sampler = dinv.sampling.DiffPIR(tr, dinv.optim.L2(), sigma=0.05, max_iter=steps, zeta=0.1, lambda_=7.0, device="cpu")
    t0 = time.time()
    out = sampler(yn, p)
    torch.randn_like = _orig
    print("cfg5 loop", time.time() - t0, flush=True)
    # the same sample path in fp64 through the oracle restatement (oracle/optim_cpu.py: diffpir): the exact-arithmetic value
    # of what the reference computes; its distance to the reference's fp32 sample is the reference's own rounding error
    from oracle import optim_cpu as OO
    from oracle import physics_cpu as O
    dt = torch.float64
    sd64 = {k: v.to(dt) for k, v in sd.items()}
    k64 = p.filter.to(dt)
    gen = g(74)
    draws = (torch.randn(1, *img, generator=gen).to(dt) for _ in range(2 * steps))
    outs64 = []

    def den64(u, s):
        o = OD.drunet(sd64, u, s)
        outs64.append(sub(o, STRIDE_TRACE).float())
        return o

    t0 = time.time()
    with torch.no_grad():
        exact = OO.diffpir(yn.to(dt), lambda v: O.downsampling_AT(v, k64, f, img),
                           lambda zz, yy, gam: O.downsampling_prox_l2(zz, yy, gam, k64, f, img),
                           den64, draws, sigma=0.05, max_iter=steps, noise_sigma=0.05)
    err = float((out.double() - exact).norm() / exact.norm())
    print("cfg5 fp64 evaluation", time.time() - t0, "reference vs fp64:", err, flush=True)
    save("cfg5_full", y=y, out=sub(out), out_exact=sub(exact).float(), out_err_vs_exact=np.float64(err),
         den_outs=torch.stack(tr.outs), den_outs_exact=torch.stack(outs64), seq=sampler.seq, stride=STRIDE,
         stride_trace=STRIDE_TRACE, drunet_seed=72, steps=steps)


CFG2_SLICES = (4, 9, 13, 18, 22, 27, 31)      # with slice 0 of cfg2_named.npz: eight slices spread over the 32 of the bench batch


def cfg2_slices():
    """BASELINE configs[1] (the headline), more slices of the bench batch through the REAL reference: `deepinv.optim.PGD`
    (50 iterations, DRUNet(2->2) with the reference's weight initialisation) on the slices CFG2_SLICES of bench.py: make_problem
    (seeded per global slice index) - cfg2_named.npz holds slice 0 only."""
    H = W = 320
    coils, iters, seed = 8, 50, 80       # (DRUNET_SEED of make_golden_r4.py)
    gm = g(0)
    maps = torch.randn(1, coils, H, W, dtype=torch.complex64, generator=gm)
    maps = maps / maps.abs().pow(2).sum(dim=1, keepdim=True).sqrt()
    from oracle import physics_cpu as OP
    mask = OP.radial_mask(H, W, 80)
    p = dinv.physics.MultiCoilMRI(mask=mask, coil_maps=maps, img_size=(2, H, W), device="cpu")
    ys = []
    for i in CFG2_SLICES:
        gi = g(1000 + i)
        x = torch.rand(1, 2, H, W, generator=gi)
        noise = torch.randn(1, 2, coils, H, W, generator=gi)
        ys.append(p.A(x) + 0.01 * noise * p.mask[:, :, None])
    y = torch.cat(ys)
    den = dinv.models.DRUNet(in_channels=2, out_channels=2, pretrained=None)
    den.load_state_dict(OD.init_state_dict(2, 2, seed=seed))
    den.eval()
    model = dinv.optim.PGD(data_fidelity=dinv.optim.L2(), prior=dinv.optim.PnP(den), stepsize=1.0, g_param=0.05, max_iter=iters,
                           early_stop=False)
    t0 = time.time()
    with torch.no_grad():
        rec = model(y, p)
    print("cfg2 slices", CFG2_SLICES, "reference PGD", time.time() - t0, "s", flush=True)
    assert torch.isfinite(rec).all()
    save("cfg2_slices", slices=np.int32(CFG2_SLICES), rec=torch.stack([sub(r) for r in rec]), rec_norm=rec.flatten(1).norm(dim=1),
         stride=STRIDE, drunet_seed=seed, iters=iters)


if __name__ == "__main__":
    torch.set_num_threads(int(os.environ.get("GOLDEN_THREADS", os.cpu_count() or 8)))
    which = sys.argv[1:] or ["cfg5", "cfg3"]
    for name in which:
        {"cfg3": cfg3, "cfg5": cfg5, "cfg2_slices": cfg2_slices}[name]()
    print("done")
