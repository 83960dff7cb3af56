"""where do the run-to-run differences of the bf16-split model under concurrent lanes sit?  (units, rows, columns of |diff| > 1e-6)"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import deepinv_amd as dinv  # noqa: E402

dev = torch.device("cuda:0")
gen = torch.Generator().manual_seed(0)
den = dinv.models.DRUNet(3, 3, pretrained=None).to(dev).eval()
den.conv_precision = "bf16split"
x = torch.rand(16, 3, 256, 256, generator=gen).to(dev)
with torch.no_grad():
    den.batch_lanes = 1
    ref = torch.cat((den(x[:8], 0.1), den(x[8:], 0.1)))
    den.batch_lanes = 2
    for it in range(30):
        o = den(x, 0.1)
        torch.cuda.synchronize()
        d = (o - ref).abs()
        if float(d.max()) > 0:
            idx = (d > 1e-6).nonzero()
            print(json.dumps({"run": it, "max": float(d.max()), "n_pixels": int(idx.shape[0]), "units": sorted(set(idx[:, 0].tolist())),
                              "rows": [int(idx[:, 2].min()), int(idx[:, 2].max())], "cols": [int(idx[:, 3].min()), int(idx[:, 3].max())],
                              "peak_at": [int(v) for v in (d == d.max()).nonzero()[0].tolist()]}), flush=True)
