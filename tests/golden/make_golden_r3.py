#!/usr/bin/env python
"""Round-3 golden vectors from the REAL reference (deepinv v0.4.1 at /root/reference through oracle/ref_shim.py) at the
NAMED shapes of BASELINE.json configs[2..4] (one unit of each config: one 512x512 image / one 12-coil 16x256x256 volume /
two 3x256x256 images), operators AND loops:

* `cfg3_named.npz`  Tomography(angles=720, img_width=512, normalize=True): operator_norm, A(x), A_adjoint(v), ramp(y),
                    fbp(y), and FBP-initialised PnP-HQS (3 iterations, prox by 3 CG iterations, DRUNet 1->1)
* `cfg4_named.npz`  MultiCoilMRI(12 coils, 16x256x256, three_d): A(x), A_adjoint(y); unfolded PGD (10 iterations) with
                    DRUNet(dim=3, nc=16..128, nb=1): reconstruction, loss and the gradients of stepsize / g_param
* `cfg5_named.npz`  Downsampling(x4, bicubic, circular) on 3x256x256: A, A_adjoint, prox_l2; DiffPIR (5 steps, DRUNet
                    3->3) with every torch.randn_like draw regenerated from a seed

Inputs are regenerated from seeds by the tests (torch CPU generators are reproducible); large outputs are stored as the
strided subsample `flat[::STRIDE]` (STRIDE is coprime with every extent, so the subsample visits every row, column,
coil and tile): the relative l2 error on the subsample is an unbiased estimate of the full one.

    python tests/golden/make_golden_r3.py [cfg3] [cfg4] [cfg5]
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
STRIDE_BIG = 101


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


def cfg3():
    W, nang = 512, 720
    t0 = time.time()
    p = dinv.physics.Tomography(angles=nang, img_width=W, circle=False, normalize=True, device="cpu", max_iter=3,
                                tol=1e-30)
    print("cfg3 init", time.time() - t0, flush=True)
    x = torch.rand(1, 1, W, W, generator=g(50))
    y = p.A(x)
    v = torch.randn(y.shape, generator=g(51))
    vadj = p.A_adjoint(v)
    ramp = p.iradon.filter(y)
    fbp = p.A_dagger(y, fbp=True)
    sd = OD.init_state_dict(1, 1, seed=52)
    den = dinv.models.DRUNet(in_channels=1, out_channels=1, pretrained=None)
    den.load_state_dict(sd)
    den.eval()
    steps, sigs = [1.0, 0.6, 0.3], [0.08, 0.05, 0.03]
    model = dinv.optim.HQS(data_fidelity=dinv.optim.L2(), prior=dinv.optim.PnP(den), stepsize=steps, g_param=sigs,
                           max_iter=3, early_stop=False, custom_init=lambda yy, pp: pp.A_dagger(yy, fbp=True))
    t0 = time.time()
    with torch.no_grad():
        rec = model(y, p)
    print("cfg3 loop", time.time() - t0, flush=True)
    save("cfg3_named", operator_norm=p.operator_norm, y=sub(y), vadj=sub(vadj), ramp=sub(ramp), fbp=sub(fbp), rec=sub(rec),
         steps=np.float32(steps), sigs=np.float32(sigs), stride=STRIDE, drunet_seed=52)


def cfg4():
    coils, vol = 12, (16, 256, 256)
    x = torch.rand(1, 2, *vol, generator=g(60))
    maps = torch.randn(1, coils, *vol, dtype=torch.complex64, generator=g(61)) / coils ** 0.5
    mask = torch.zeros(*vol)
    mask[..., ::4] = 1
    mask[..., 118:138] = 1
    p = dinv.physics.MultiCoilMRI(mask=mask, coil_maps=maps, img_size=(2, *vol), three_d=True)
    y = p.A(x)
    yadj = p.A_adjoint(y)
    torch.manual_seed(62)
    den = dinv.models.DRUNet(in_channels=2, out_channels=2, nc=(16, 32, 64, 128), nb=1, pretrained=None, dim=3)
    w_probe = den.m_head.weight.detach().reshape(-1)[:16].clone()
    model = dinv.unfolded.unfolded_builder("PGD", data_fidelity=dinv.optim.L2(), prior=dinv.optim.PnP(den),
                                           params_algo={"stepsize": 1.0, "g_param": 0.05, "lambda": 1.0}, max_iter=10,
                                           trainable_params=["stepsize", "g_param"])
    t0 = time.time()
    rec = model(y, p)
    loss = (rec - x).pow(2).mean()
    loss.backward()
    print("cfg4 loop", time.time() - t0, float(loss), flush=True)
    gp = dict(model.named_parameters())
    save("cfg4_named", y=sub(y, STRIDE_BIG), yadj=sub(yadj), rec=sub(rec), loss=loss.detach(),
         grad_stepsize=gp["init_params_algo.stepsize.0"].grad, grad_g_param=gp["init_params_algo.g_param.0"].grad,
         grad_head=sub(den.m_head.weight.grad, 1), w_probe=w_probe, stride=STRIDE, stride_y=STRIDE_BIG, drunet_seed=62,
         max_iter=10)


def cfg5():
    img, f, B = (3, 256, 256), 4, 2
    x = torch.rand(B, *img, generator=g(70))
    p = dinv.physics.Downsampling(img_size=img, filter="bicubic", factor=f, padding="circular",
                                  noise_model=dinv.physics.GaussianNoise(0.05))
    y = p.A(x)
    z = torch.rand(B, *img, generator=g(71))
    yadj = p.A_adjoint(y)
    prox = p.prox_l2(z, y, 0.7)
    sd = OD.init_state_dict(3, 3, seed=72)
    den = dinv.models.DRUNet(in_channels=3, out_channels=3, pretrained=None)
    den.load_state_dict(sd)
    den.eval()
    yn = y[:1] + 0.05 * torch.randn(1, 3, 64, 64, generator=g(73))
    gen = g(74)
    _orig = torch.randn_like
    torch.randn_like = lambda t, **kw: torch.randn(t.shape, generator=gen)
    sampler = dinv.sampling.DiffPIR(den, dinv.optim.L2(), sigma=0.05, max_iter=5, zeta=0.1, lambda_=7.0, device="cpu")
    t0 = time.time()
    out = sampler(yn, p)
    torch.randn_like = _orig
    print("cfg5 loop", time.time() - t0, flush=True)
    # the same sample path in fp64 through the oracle restatement (oracle/optim_cpu.py: diffpir), i.e. the exact-arithmetic
    # value of what the reference computes: its distance to the reference's fp32 sample is the reference's own rounding
    # error (dominated by the gamma = 7e5 prox of the first step)
    from oracle import optim_cpu as OO
    from oracle import physics_cpu as O
    dt = torch.float64
    sd64 = {k: v.to(dt) for k, v in sd.items()}
    k64 = p.filter.to(dt)
    gen = g(74)
    draws = (torch.randn(1, *img, generator=gen).to(dt) for _ in range(64))
    with torch.no_grad():
        exact = OO.diffpir(yn.to(dt), lambda v: O.downsampling_AT(v, k64, f, img),
                           lambda zz, yy, gam: O.downsampling_prox_l2(zz, yy, gam, k64, f, img),
                           lambda u, s: OD.drunet(sd64, u, s), draws, sigma=0.05, max_iter=5, noise_sigma=0.05)
    err = float((out.double() - exact).norm() / exact.norm())
    print("cfg5 reference vs fp64 evaluation", err, flush=True)
    save("cfg5_named", y=y, yadj=sub(yadj), prox=sub(prox), out=sub(out), out_exact=sub(exact).float(),
         out_err_vs_exact=np.float64(err), seq=sampler.seq, rhos=sampler.rhos, sigmas=sampler.sigmas, stride=STRIDE,
         drunet_seed=72)


if __name__ == "__main__":
    torch.set_num_threads(os.cpu_count() or 8)
    which = sys.argv[1:] or ["cfg5", "cfg4", "cfg3"]
    for name in which:
        {"cfg3": cfg3, "cfg4": cfg4, "cfg5": cfg5}[name]()
    print("done")
