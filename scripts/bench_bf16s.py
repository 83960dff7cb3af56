"""Time the bf16-split ResBlock conv at the four DRUNet levels (B images of 320x320 at level 0).
Usage: [DINV_BF16S_WAVES=4] python scripts/bench_bf16s.py [B]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepinv_amd.hip import drunet as K  # noqa: E402
from bench_ops import timeit  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda:0")
tot = 0.0
for lvl, c in enumerate((64, 128, 256, 512)):
    H = 320 >> lvl
    g = K.geom(B, H, H)
    x, y, r = K.alloc(g, c, dev), K.alloc(g, c, dev), K.alloc(g, c, dev)
    x.normal_()
    r.normal_()
    w = torch.randn(c, c, 3, 3, device=dev) / (3 * c ** 0.5)
    ws = K.pack_bf16s_weight(w)
    fl = 2.0 * 9 * c * c * B * H * H
    t1 = timeit(lambda: K.conv3x3_bf16s(g, x, ws, c, c, y, relu=True), iters=20, warmup=3)
    t2 = timeit(lambda: K.conv3x3_bf16s(g, x, ws, c, c, y, res1=r), iters=20, warmup=3)
    n = 8 if lvl == 3 else 16
    tot += n / 2 * (t1 + t2)
    print(json.dumps({"lvl": lvl, "B": B, "relu_ms": round(t1 * 1e3, 4), "res_ms": round(t2 * 1e3, 4),
                      "executed_TF": round(3 * fl / t1 / 1e12, 1), "waves": os.environ.get("DINV_BF16S_WAVES", "8")}))
print(json.dumps({"resblock_convs_ms_per_drunet": round(tot * 1e3, 2)}))
for lvl, (ci, co) in enumerate(((64, 128), (128, 256), (256, 512))):
    H = 320 >> lvl
    gi, go = K.geom(B, H, H), K.geom(B, H // 2, H // 2)
    x, y = K.alloc(gi, ci, dev), K.alloc(go, co, dev)
    x.normal_()
    w = torch.randn(co, ci, 2, 2, device=dev) / (2 * ci ** 0.5)
    wf, wb = K.pack_down_weight(w), K.pack_down_bf16s_weight(w)
    t1 = timeit(lambda: K.down2x2(gi, go, x, wf, ci, co, y), iters=20, warmup=3)
    t2 = timeit(lambda: K.down2x2_bf16s(gi, go, x, wb, ci, co, y), iters=20, warmup=3)
    print(json.dumps({"down": lvl, "B": B, "fp32_ms": round(t1 * 1e3, 4), "bf16s_ms": round(t2 * 1e3, 4),
                      "hbm_floor_ms": round((x.numel() + y.numel()) * 4 / 5e12 * 1e3, 4)}))
for lvl, (ci, co) in enumerate(((128, 64), (256, 128), (512, 256))):
    H = 160 >> lvl
    gi, go = K.geom(B, H, H), K.geom(B, 2 * H, 2 * H)
    x, x2, y = K.alloc(gi, ci, dev), K.alloc(gi, ci, dev), K.alloc(go, co, dev)
    x.normal_()
    x2.normal_()
    w = torch.randn(ci, co, 2, 2, device=dev) / ci ** 0.5
    wf, wb = K.pack_up_weight(w), K.pack_up_bf16s_weight(w)
    t1 = timeit(lambda: K.up2x2(gi, go, x, x2, wf, ci, co, y), iters=20, warmup=3)
    t2 = timeit(lambda: K.up2x2_bf16s(gi, go, x, x2, wb, ci, co, y), iters=20, warmup=3)
    print(json.dumps({"up": lvl, "B": B, "fp32_ms": round(t1 * 1e3, 4), "bf16s_ms": round(t2 * 1e3, 4),
                      "hbm_floor_ms": round((2 * x.numel() + y.numel()) * 4 / 5e12 * 1e3, 4)}))
