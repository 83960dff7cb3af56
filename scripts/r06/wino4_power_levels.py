"""Is every DRUNet level of the F(4x4,3x3) kernel at the package power limit?  (VERDICT r5 "next" #4: the power-cap argument of
DESIGN 3.2 was only shown for the MFMA-dense regime.)  Per level (320^2 x 64 ch ... 40^2 x 512 ch) and epilogue form (ReLU /
residual), `seconds` of back-to-back launches with rocm-smi sampled beside them by bench.py's PowerSampler: ms per launch sustained,
median socket power, median shader clock.   python scripts/r06/wino4_power_levels.py [batch] [seconds] [bf16x3]"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bench import PowerSampler  # noqa: E402
from deepinv_amd.hip import drunet as K  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
SECONDS = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
BF3 = len(sys.argv) > 3 and sys.argv[3] == "bf16x3"
dev = torch.device("cuda:0")
print(json.dumps({"batch": B, "seconds_per_row": SECONDS, "form": "bf16x3" if BF3 else "fp32"}), flush=True)
for level, (side, ch) in enumerate(((320, 64), (160, 128), (80, 256), (40, 512))):
    g = torch.Generator().manual_seed(level)
    geo = K.geom(B, side, side)
    x = torch.randn(B, ch, side, side, generator=g).relu_().to(dev)       # post-ReLU activations: half zeros, like the network's
    r = torch.randn(B, ch, side, side, generator=g).to(dev)
    w = (torch.randn(ch, ch, 3, 3, generator=g) / (3.0 * ch ** 0.5)).to(dev)

    def to_act(t):
        a = K.alloc(geo, ch, dev)
        a[:, geo.sl:geo.sl + geo.np].view(-1, B, geo.hp, geo.wp, 8)[:, :, 1:side + 1, 1:side + 1] = t.view(B, -1, 8, side, side).permute(1, 0, 3, 4, 2)
        return a

    xa, ra, ya = to_act(x), to_act(r), K.alloc(geo, ch, dev)
    wp = K.pack_winograd4_bf16x3_weight(w) if BF3 else K.pack_winograd4_weight(w)
    ws = K.winograd4_workspace(dev)
    fn = K.conv3x3_winograd4_bf16x3 if BF3 else K.conv3x3_winograd4
    del x, r
    for mode in ("relu", "res"):
        kw = dict(res1=ra if mode == "res" else None, relu=mode == "relu", workspace=ws)
        for _ in range(20):
            fn(geo, xa, wp, ch, ch, ya, **kw)
        torch.cuda.synchronize()
        # burst of 30 launches (what a harness measures), then the sustained run
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        time.sleep(1.0)
        e0.record()
        for _ in range(30):
            fn(geo, xa, wp, ch, ch, ya, **kw)
        e1.record()
        torch.cuda.synchronize()
        burst = e0.elapsed_time(e1) / 30
        n = max(int(SECONDS * 1e3 / burst), 100)
        sampler = PowerSampler(0, period=0.25)
        e0.record()
        for _ in range(n):
            fn(geo, xa, wp, ch, ch, ya, **kw)
        e1.record()
        torch.cuda.synchronize()
        pkg = sampler.stop()
        ms = e0.elapsed_time(e1) / n
        flops = 2.0 * 36 * ch * ch * B * side * side / 16
        print(json.dumps({"level": level, "side": side, "channels": ch, "mode": mode, "launches": n, "ms_sustained": round(ms, 4),
                          "ms_burst30": round(burst, 4), "executed_TFLOPs": round(flops / ms / 1e9, 1),
                          "frac_fp32_mfma_peak": round(flops / ms / 1e9 / 157.3, 3), "package": pkg}), flush=True)
    del xa, ra, ya
