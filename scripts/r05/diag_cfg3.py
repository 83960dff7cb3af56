"""diagnostic: configs[2] full length on the GPU against tests/golden/cfg3_full.npz, per iteration: CG iterations of the prox
(host check every iteration, so the count is exact) and the error of the denoiser output"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import deepinv_amd as dinv
from deepinv_amd.optim import linear
from oracle import drunet_cpu as OD
linear.CG_CHECK_EVERY = 1
dev = torch.device("cuda:0")
d = {k: torch.from_numpy(np.asarray(v)) for k, v in np.load(os.path.join(ROOT, "tests/golden/cfg3_full.npz")).items()}
stt, iters = int(d["stride_trace"]), int(d["iters"])
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
x = torch.rand(1, 1, 512, 512, generator=torch.Generator().manual_seed(50)).expand(B, 1, 512, 512).contiguous().to(dev)
p = dinv.physics.Tomography(angles=720, img_width=512, circle=False, normalize=True, device=dev)
print("operator norm", float(p.operator_norm), float(d["operator_norm"]))
y = p.A(x)
den = dinv.models.DRUNet(1, 1, pretrained=None).to(dev).eval()
den.load_state_dict(OD.init_state_dict(1, 1, seed=int(d["drunet_seed"])))
trace, counts, cnt = [], [], [0]
den.register_forward_hook(lambda m, i, o: trace.append(o[:1].detach().reshape(-1)[::stt].cpu()))
ata, prox = p.A_adjoint_A, p.prox_l2
def c_ata(v, **kw):
    cnt[0] += 1
    return ata(v, **kw)
def c_prox(*a, **kw):
    cnt[0] = 0
    o = prox(*a, **kw)
    counts.append(cnt[0])
    return o
p.A_adjoint_A, p.prox_l2 = c_ata, c_prox
model = dinv.optim.HQS(data_fidelity=dinv.optim.L2(), prior=dinv.optim.PnP(den), stepsize=[float(v) for v in d["steps"]],
                       g_param=[float(v) for v in d["sigs"]], max_iter=iters, early_stop=False,
                       custom_init=lambda yy, pp: pp.A_dagger(yy, fbp=True))
with torch.no_grad():
    rec = model(y, p)
rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
print("final", rel(rec[:1].reshape(-1)[::int(d["stride"])].cpu(), d["rec"]))
print("ata per prox (gpu):", counts)
print("ata per prox (ref):", d["n_ata"].tolist())
print("trace err:", [f"{rel(a, b):.1e}" for a, b in zip(trace, d["den_outs"])])
