#!/usr/bin/env python
"""Round-4 golden vectors from the REAL reference (deepinv v0.4.1 at /root/reference through oracle/ref_shim.py) at the
HEADLINE shape, BASELINE.json configs[1]: one slice of bench.py's own seeded problem (8 coils, 320x320, 80-spoke radial
mask) through the reference's `MultiCoilMRI` (deepinv/physics/mri.py:254-324) and its `PGD` optimizer
(deepinv/optim/optimizers.py:1596-1734) with the DRUNet(2->2) prior, 50 iterations:

* `cfg2_named.npz`  A(x), A_adjoint(y) and the 50-iteration PnP-PGD reconstruction of slice 0 of the bench batch, twice:
                    with the reference's weight initialisation (orthogonal, gain 0.2) and with the 56 ResBlock
                    convolutions at gain `RES_GAIN` (the ResBlock branches are then O(1) against the identity path, so
                    the end-to-end figure is a statement about the ResBlock kernels).
                    Also: seconds per 50-iteration slice of the real reference and of the oracle port on the same cores
                    (backs `cpu_baseline.kind = "port"` of bench.py).

Inputs are regenerated from seeds by the test (same generators as bench.py: make_problem); large outputs are stored as
the strided subsample flat[::STRIDE].

    python tests/golden/make_golden_r4.py            # writes cfg2_named.npz (about 4 minutes on 8 cores)
    python tests/golden/make_golden_r4.py --probe    # per-iteration norms of the unit-gain loop through the port
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import drunet_cpu as OD  # noqa: E402
from oracle import optim_cpu as OO  # noqa: E402
from oracle import physics_cpu as OP  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
STRIDE = 7
H = W = 320
COILS = 8
ITERS = 50
DRUNET_SEED = 80
RES_GAIN = 1.0


def radial_mask(Hh, Ww, n_spokes):
    """the bench's mask (deepinv_amd/utils/synthetic.py: radial_mask), restated so that this script needs no product import"""
    import math
    mask = torch.zeros(Hh, Ww)
    cy, cx = Hh // 2, Ww // 2
    L = int(math.ceil(math.hypot(Hh, Ww)))
    t = torch.arange(-L, L + 1, dtype=torch.float64)
    for k in range(n_spokes):
        a = math.pi * k / n_spokes
        yy = torch.round(cy + t * math.sin(a)).long()
        xx = torch.round(cx + t * math.cos(a)).long()
        ok = (yy >= 0) & (yy < Hh) & (xx >= 0) & (xx < Ww)
        mask[yy[ok], xx[ok]] = 1.0
    return mask


def problem():
    """slice 0 of bench.py: make_problem (same generators, same order of draws)"""
    g = torch.Generator().manual_seed(0)
    maps = torch.randn(1, COILS, H, W, dtype=torch.complex64, generator=g)
    maps = maps / maps.abs().pow(2).sum(dim=1, keepdim=True).sqrt()
    mask = radial_mask(H, W, 80)
    gi = torch.Generator().manual_seed(1000)
    x = torch.rand(1, 2, H, W, generator=gi)
    noise = torch.randn(1, 2, COILS, H, W, generator=gi)
    return maps, mask, x, noise


def sub(t):
    return t.detach().reshape(-1)[::STRIDE].clone()


def port_pgd(sd, maps, mask, y, iters, trace=None):
    A = lambda v: OP.multicoil_A(v, maps, mask)
    AT = lambda v: OP.multicoil_AT(v, maps, mask)

    def den(u, s):
        o = OD.drunet(sd, u, s)
        if trace is not None:
            trace.append(float(o.norm()))
        return o

    with torch.no_grad():
        return OO.pnp_pgd(y, A, AT, den, max_iter=iters)


def probe():
    maps, mask, x, noise = problem()
    m2 = OP.check_mask(mask, False)
    y = OP.multicoil_A(x, maps, mask) + 0.01 * noise * m2[:, :, None]
    for gain in (0.2, 0.5, 1.0):
        sd = OD.init_state_dict(2, 2, seed=DRUNET_SEED, res_gain=None if gain == 0.2 else gain)
        tr = []
        t0 = time.time()
        port_pgd(sd, maps, mask, y, 12, tr)
        print("gain", gain, "norms", ["%.3g" % v for v in tr], "%.1fs" % (time.time() - t0), flush=True)


def main():
    from oracle.ref_shim import import_reference
    dinv = import_reference()
    torch.set_num_threads(os.cpu_count() or 1)
    maps, mask, x, noise = problem()
    p = dinv.physics.MultiCoilMRI(mask=mask, coil_maps=maps, img_size=(2, H, W), device="cpu")
    y0 = p.A(x)
    y = y0 + 0.01 * noise * p.mask[:, :, None]
    yadj = p.A_adjoint(y)
    out = {"y": sub(y0), "yadj": sub(yadj), "stride": STRIDE, "drunet_seed": DRUNET_SEED, "res_gain": RES_GAIN, "iters": ITERS}
    for tag, gain in (("", None), ("_gain", RES_GAIN)):
        sd = OD.init_state_dict(2, 2, seed=DRUNET_SEED, res_gain=gain)
        den = dinv.models.DRUNet(in_channels=2, out_channels=2, pretrained=None)
        den.load_state_dict(sd)
        den.eval()
        model = dinv.optim.PGD(data_fidelity=dinv.optim.L2(), prior=dinv.optim.PnP(den), stepsize=1.0, g_param=0.05,
                               max_iter=ITERS, early_stop=False)
        t0 = time.time()
        with torch.no_grad():
            rec = model(y, p)
        t_ref = time.time() - t0
        t0 = time.time()
        rec_port = port_pgd(sd, maps, mask, y, ITERS)
        t_port = time.time() - t0
        d = float((rec.double() - rec_port.double()).norm() / rec.double().norm())
        print(f"cfg2{tag}: reference {t_ref:.1f} s, port {t_port:.1f} s, port vs reference {d:.3e}, |rec| {float(rec.norm()):.4g}",
              flush=True)
        assert torch.isfinite(rec).all()
        out["rec" + tag] = sub(rec)
        out["rec_norm" + tag] = float(rec.norm())
        out["seconds_reference" + tag] = t_ref
        out["seconds_port" + tag] = t_port
        out["port_vs_reference" + tag] = d
    out["threads"] = torch.get_num_threads()
    np.savez_compressed(os.path.join(OUT, "cfg2_named.npz"),
                        **{k: (v.numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in out.items()})
    print("cfg2_named", {k: np.asarray(v).shape for k, v in out.items()}, flush=True)


if __name__ == "__main__":
    probe() if "--probe" in sys.argv else main()
