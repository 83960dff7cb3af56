"""DRUNet end to end with the experimental bf16-split convolutions (DINV_CONV_BF16X3=2|3): error vs the CPU oracle and time"""
import os, sys, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import deepinv_amd as dinv
from oracle import drunet_cpu as OD

dev = torch.device("cuda:0")
sd = OD.init_state_dict(2, 2, seed=1)
x = torch.rand(2, 2, 64, 96, generator=torch.Generator().manual_seed(0))
ref = OD.drunet(sd, x, 0.05)
model = dinv.models.DRUNet(2, 2, pretrained=None).to(dev).eval()
model.load_state_dict(sd)
with torch.no_grad():
    out = model(x.to(dev), 0.05).cpu()
err = float((out.double() - ref.double()).norm() / ref.double().norm())
xb = torch.rand(32, 2, 320, 320, device=dev)
with torch.no_grad():
    for _ in range(2): model(xb, 0.05)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(4): model(xb, 0.05)
    e1.record(); torch.cuda.synchronize()
print(json.dumps({"mode": os.environ.get("DINV_CONV_BF16X3", "winograd fp32"), "drunet_rel_err_vs_oracle": err,
                  "forward_ms_B32": e0.elapsed_time(e1) / 4}))
