"""bf16-split model under concurrent lanes: column histogram of the differing pixels (mod 52 = the tail kernel's strip width at 256 columns), and the
same with the tail convolution on the MFMA kernel instead of tail3x3_shift_kernel."""
import collections
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import deepinv_amd as dinv  # noqa: E402

dev = torch.device("cuda:0")
gen = torch.Generator().manual_seed(0)
x = torch.rand(16, 3, 256, 256, generator=gen).to(dev)
for variant in ("tail3x3_shift_kernel", "tail on conv3x3_kernel"):
    den = dinv.models.DRUNet(3, 3, pretrained=None).to(dev).eval()
    den.conv_precision = "bf16split"
    with torch.no_grad():
        den.batch_lanes = 1
        den(x[:2], 0.1)
        if variant != "tail3x3_shift_kernel":
            den._engine["tail_valu"] = None
        ref = torch.cat((den(x[:8], 0.1), den(x[8:], 0.1)))
        den.batch_lanes = 2
        hist, nbad, chans = collections.Counter(), 0, collections.Counter()
        for it in range(30):
            o = den(x, 0.1)
            torch.cuda.synchronize()
            d = (o - ref).abs()
            if float(d.max()) > 0:
                nbad += 1
                idx = (d > 1e-6).nonzero()
                hist.update((idx[:, 3] % 52).tolist())
                chans.update(idx[:, 1].tolist())
    print(json.dumps({"variant": variant, "runs_that_differ_of_30": nbad, "columns_mod_52": dict(sorted(hist.items())), "channels": dict(chans)}), flush=True)
