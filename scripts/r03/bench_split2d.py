"""The bf16-split 3x3 conv (csrc/drunet_split2d.hip) at the four DRUNet levels on one MI355X: conv1 = fp32 in -> pre-split ReLU
out, conv2 = pre-split in + fp32 residual -> fp32 out, with the automatic tile size and with 128- / 256-pixel workgroups forced.
Usage: python scripts/r03/bench_split2d.py [B]"""
import ctypes
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepinv_amd.hip import check, ptr, stream_ptr  # noqa: E402
from deepinv_amd.hip import drunet as K  # noqa: E402
from bench_ops import timeit  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda:0")
tot = {0: 0.0, 1: 0.0, 2: 0.0}
for lvl, c in enumerate((64, 128, 256, 512)):
    H = 320 >> lvl
    g = K.geom(B, H, H)
    x, y, r, t = K.alloc(g, c, dev), K.alloc(g, c, dev), K.alloc(g, c, dev), K.alloc(g, c, dev)
    x[:, g.sl:g.sl + g.np].view(-1, B, g.hp, g.wp, 8)[:, :, 1:H + 1, 1:H + 1].normal_()
    r[:, g.sl:g.sl + g.np].view(-1, B, g.hp, g.wp, 8)[:, :, 1:H + 1, 1:H + 1].normal_()
    w = torch.randn(c, c, 3, 3, device=dev) / (3 * c ** 0.5)
    w2 = K.pack_split2d_weight(w)
    fl = 2.0 * 9 * c * c * B * H * H

    def raw(xx, yy, res, flags):
        check(K._l().dinv_conv3x3_split(ctypes.byref(g), ptr(xx), ptr(w2), c, c, ptr(yy), ptr(res), flags, stream_ptr(dev)))

    row = {"lvl": lvl, "B": B}
    n = 8 if lvl == 3 else 16
    for tile, name in ((0, "auto"), (1, "128px"), (2, "256px")):
        t1 = timeit(lambda: raw(x, t, None, 4 | 2 | (tile << 8)), iters=30, warmup=3)
        t2 = timeit(lambda: raw(t, y, r, 1 | (tile << 8)), iters=30, warmup=3)
        row[f"conv1_{name}_ms"] = round(t1 * 1e3, 4)
        row[f"conv2_{name}_ms"] = round(t2 * 1e3, 4)
        tot[tile] += n / 2 * (t1 + t2)
    row["conv2_auto_direct_TF"] = round(fl / (row["conv2_auto_ms"] * 1e-3) / 1e12, 1)
    print(json.dumps(row), flush=True)
print(json.dumps({"B": B, "resblock_convs_ms_per_drunet": {"auto": round(tot[0] * 1e3, 2), "128px": round(tot[1] * 1e3, 2),
                                                            "256px": round(tot[2] * 1e3, 2)}}))
