"""is the bf16-split conv power-limited?  same launch on random / zero activations and weights (DVFS gives zero data a higher clock).
The run recorded in profiles/r03_zero_data_test.jsonl also compared three software pipelines of the K step that were selectable
through flag bits 10-11 at the time (mode 0 / 1 / 2; only mode 2 is in the tree now)"""
import json, os, sys, ctypes
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepinv_amd.hip import drunet as K, check, ptr, stream_ptr
from bench_ops import timeit
dev = torch.device("cuda:0")
B = 32
for lvl in (0, 1, 2):
    c, H = 64 << lvl, 320 >> lvl
    g = K.geom(B, H, H)
    x, y, t = K.alloc(g, c, dev), K.alloc(g, c, dev), K.alloc(g, c, dev)
    w = torch.randn(c, c, 3, 3, device=dev) / (3 * c ** 0.5)
    for data in ("random", "zero_x", "zero_all"):
        if data == "random":
            x[:, g.sl:g.sl + g.np].view(-1, B, g.hp, g.wp, 8)[:, :, 1:H + 1, 1:H + 1].normal_()
        else:
            x.zero_()
        w2 = K.pack_split2d_weight(w if data != "zero_all" else torch.zeros_like(w))
        for mode in (2,):          # the committed pipeline
            def raw(xx, yy, flags):
                check(K._l().dinv_conv3x3_split(ctypes.byref(g), ptr(xx), ptr(w2), c, c, ptr(yy), None, flags, stream_ptr(dev)))
            raw(x, t, 4 | 2)
            ms = timeit(lambda: raw(t, y, 1), iters=30, warmup=5) * 1e3
            print(json.dumps({"lvl": lvl, "data": data, "mode": mode, "conv2_ms": round(ms, 4),
                              "executed_PF": round(3 * 2 * 9 * c * c * B * H * H / ms / 1e12, 3)}), flush=True)
