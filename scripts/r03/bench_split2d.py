"""A/B on one MI355X: the round-2 bf16-split conv (1-D row segments) vs the 2-D-tile kernel (csrc/drunet_split2d.hip) at the
four DRUNet levels; conv1 = fp32 in -> pre-split ReLU out, conv2 = pre-split in + fp32 residual -> fp32 out.
Usage: python scripts/r03/bench_split2d.py [B]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepinv_amd.hip import drunet as K  # noqa: E402
from bench_ops import timeit  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda:0")
tot_old = tot_new = 0.0
for lvl, c in enumerate((64, 128, 256, 512)):
    H = 320 >> lvl
    g = K.geom(B, H, H)
    x, y, r, t = K.alloc(g, c, dev), K.alloc(g, c, dev), K.alloc(g, c, dev), K.alloc(g, c, dev)
    x[:, g.sl:g.sl + g.np].view(-1, B, g.hp, g.wp, 8)[:, :, 1:H + 1, 1:H + 1].normal_()
    r[:, g.sl:g.sl + g.np].view(-1, B, g.hp, g.wp, 8)[:, :, 1:H + 1, 1:H + 1].normal_()
    w = torch.randn(c, c, 3, 3, device=dev) / (3 * c ** 0.5)
    fl = 2.0 * 9 * c * c * B * H * H
    row = {"lvl": lvl, "B": B}
    if hasattr(K, "conv3x3_bf16s"):
        ws = K.pack_bf16s_weight(w)
        o1 = timeit(lambda: K.conv3x3_bf16s(g, x, ws, c, c, y, relu=True), iters=20, warmup=3)
        o2 = timeit(lambda: K.conv3x3_bf16s(g, x, ws, c, c, y, res1=r), iters=20, warmup=3)
        row.update(old_relu_ms=round(o1 * 1e3, 4), old_res_ms=round(o2 * 1e3, 4))
        y_old = y.clone()
    w2 = K.pack_split2d_weight(w)
    n1 = timeit(lambda: K.conv3x3_split(g, x, w2, c, c, t, relu=True, y_presplit=True), iters=20, warmup=3)
    n2 = timeit(lambda: K.conv3x3_split(g, t, w2, c, c, y, res1=r, x_presplit=True), iters=20, warmup=3)
    n3 = timeit(lambda: K.conv3x3_split(g, x, w2, c, c, y, res1=r), iters=20, warmup=3)
    import ctypes
    from deepinv_amd.hip import check, ptr, stream_ptr
    def raw(xx, yy, res, flags):
        check(K._l().dinv_conv3x3_split(ctypes.byref(g), ptr(xx), ptr(w2), c, c, ptr(yy), ptr(res), flags, stream_ptr(dev)))
    s1 = timeit(lambda: raw(x, t, None, 4 | 2 | 0x400), iters=20, warmup=3)
    s2 = timeit(lambda: raw(t, y, r, 1 | 0x400), iters=20, warmup=3)
    row.update(sact_conv1_ms=round(s1 * 1e3, 4), sact_conv2_ms=round(s2 * 1e3, 4))
    s1 = timeit(lambda: raw(x, t, None, 4 | 2 | 0x800), iters=20, warmup=3)
    s2 = timeit(lambda: raw(t, y, r, 1 | 0x800), iters=20, warmup=3)
    row.update(step_conv1_ms=round(s1 * 1e3, 4), step_conv2_ms=round(s2 * 1e3, 4))
    row.update(conv1_ms=round(n1 * 1e3, 4), conv2_ms=round(n2 * 1e3, 4), f32_res_ms=round(n3 * 1e3, 4),
               conv2_direct_TF=round(fl / n2 / 1e12, 1), conv1_direct_TF=round(fl / n1 / 1e12, 1))
    if hasattr(K, "conv3x3_bf16s"):
        row["f32_res_vs_old_maxdiff"] = float((y - y_old).abs().max())
        n = 8 if lvl == 3 else 16
        tot_old += n / 2 * (o1 + o2)
    n = 8 if lvl == 3 else 16
    tot_new += n / 2 * (n1 + n2)
    print(json.dumps(row), flush=True)
print(json.dumps({"B": B, "resblock_convs_ms_per_drunet_old": round(tot_old * 1e3, 2), "new": round(tot_new * 1e3, 2)}))
