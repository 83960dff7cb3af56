"""BASELINE config 4's loop on one MI355X: unfolded PGD x10 + DRUNet(dim=3, nc=16..128, nb=1) training step on two 12-coil
16x256x256 volumes (for rocprofv3 --kernel-trace --stats); prints the step time"""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import deepinv_amd as dinv
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
B, coils, vol = 2, 12, (16, 256, 256)
x = torch.rand(B, 2, *vol, generator=g).to(dev)
maps = (torch.randn(1, coils, *vol, dtype=torch.complex64, generator=g) / coils ** 0.5).to(dev)
mask = torch.zeros(*vol); mask[..., ::4] = 1; mask[..., 118:138] = 1
phys = dinv.physics.MultiCoilMRI(mask=mask.to(dev), coil_maps=maps, img_size=(2, *vol), three_d=True, device=dev)
y = phys.A(x)
torch.manual_seed(0)
den = dinv.models.DRUNet(2, 2, nc=(16, 32, 64, 128), nb=1, pretrained=None, dim=3).to(dev)
if len(sys.argv) > 1:
    den.train_forward_precision = sys.argv[1]
net = dinv.unfolded.unfolded_builder("PGD", data_fidelity=dinv.optim.L2(), prior=dinv.optim.PnP(den),
                                     params_algo={"stepsize": 1.0, "g_param": 0.05, "lambda": 1.0}, max_iter=10,
                                     trainable_params=["stepsize", "g_param"], device=dev).to(dev)
def step():
    net.zero_grad()
    (net(y, phys) - x).pow(2).mean().backward()
step(); torch.cuda.synchronize()
t0 = time.perf_counter()
n = int(os.environ.get("N", "2"))
for _ in range(n):
    step()
torch.cuda.synchronize()
print(json.dumps({"cfg4_train_step_ms": (time.perf_counter() - t0) / n * 1e3, "forward": den.train_forward_precision}))
