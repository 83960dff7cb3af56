"""bisect the non-reproducible bf16-split model under concurrent lanes by kernel family: A = ResBlock convs on the direct fp32 kernel, stride-2
layers bf16-split; B = ResBlock convs wsplit, stride-2 layers fp32; C = everything bf16-split (the failing set); D = wsplit only in ONE level."""
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


def strip(pk):      # ResBlock conv packs without the bf16-split forms -> direct fp32 kernel
    return (pk[0], pk[1], None, None, None, None, None)


def variant(name, edit):
    den = dinv.models.DRUNet(3, 3, pretrained=None).to(dev).eval()
    den.conv_precision = "bf16split"
    with torch.no_grad():
        den.batch_lanes = 1
        den(x[:2], 0.1)                      # builds the engine
        edit(den._engine)
        ref = torch.cat((den(x[:8], 0.1), den(x[8:], 0.1)))
        den.batch_lanes = 2
        outs = [den(x, 0.1) for _ in range(25)]
    torch.cuda.synchronize()
    diffs = [float((o - ref).abs().max()) for o in outs]
    print(json.dumps({"variant": name, "distinct": len(set(diffs)), "max_abs_diff": max(diffs)}), flush=True)


LEVELS = {"m_down1": 0, "m_down2": 1, "m_down3": 2, "m_body": 3, "m_up3": 2, "m_up2": 1, "m_up1": 0}


def res_fp32(e, only_levels=None):
    for name, lvl in LEVELS.items():
        if only_levels is None or lvl in only_levels:
            e[name] = [(strip(a), strip(b)) for a, b in e[name]]


def stride_fp32(e):
    for name in ("m_down1", "m_down2", "m_down3", "m_up3", "m_up2", "m_up1"):
        e[name + "_sb"] = None


variant("C all bf16-split", lambda e: None)
variant("A ResBlocks direct fp32, stride-2 bf16s", lambda e: res_fp32(e))
variant("B ResBlocks wsplit, stride-2 fp32", stride_fp32)
for keep in range(4):
    variant(f"D wsplit only at level {keep}, stride-2 fp32", lambda e, keep=keep: (stride_fp32(e), res_fp32(e, {0, 1, 2, 3} - {keep})))
