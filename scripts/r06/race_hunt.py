"""Which kernel of the bf16-split DRUNet path gives run-to-run different results when TWO launch sequences run concurrently on two
streams (batch lanes)?  Every candidate kernel is run alone on stream A (reference), then 30 times on streams A and B at once on
separate buffers; outputs are compared bit for bit.  Also the whole model."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import deepinv_amd as dinv  # noqa: E402
from deepinv_amd.hip import drunet as K  # noqa: E402

dev = torch.device("cuda:0")
sA, sB = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
gen = torch.Generator().manual_seed(0)
B = 8


def act(g, c, fill=True):
    a = K.alloc(g, c, dev)
    if fill:
        H, W = g.height, g.width
        t = torch.randn(B, c, H, W, generator=gen).relu_().to(dev)
        a[:, g.sl:g.sl + g.np].view(-1, B, g.hp, g.wp, 8)[:, :, 1:H + 1, 1:W + 1] = t.view(B, -1, 8, H, W).permute(1, 0, 3, 4, 2)
    return a


def hunt(name, make):
    """make() -> (launch(stream_index) -> output tensor) for two independent problem instances"""
    runs = [make(), make()]
    refs = []
    for i, s in enumerate((sA, sB)):
        with torch.cuda.stream(s):
            refs.append(runs[i]().clone())
        torch.cuda.synchronize()
    bad = 0
    for it in range(30):
        outs = []
        for i, s in enumerate((sA, sB)):
            with torch.cuda.stream(s):
                outs.append(runs[i]())
        torch.cuda.synchronize()
        bad += sum(0 if torch.equal(o, r) else 1 for o, r in zip(outs, refs))
    print(json.dumps({"kernel": name, "mismatching_outputs_of_60": bad}), flush=True)


for lvl, (side, c) in enumerate(((256, 64), (128, 128), (64, 256), (32, 512))):
    g = K.geom(B, side, side)
    w = (torch.randn(c, c, 3, 3, generator=gen) / (3.0 * c ** 0.5)).to(dev)
    wws = K.pack_wsplit_weight(w)
    s2d = K.pack_split2d_weight(w)
    w4 = K.pack_winograd4_weight(w)

    def mk_wsplit(relu):
        def make():
            x, r, y = act(g, c), act(g, c), act(g, c, False)
            return lambda: (K.conv3x3_wsplit(g, x, wws, c, c, y, res1=None if relu else r, relu=relu), y)[1]
        return make

    hunt(f"wsplit relu L{lvl}", mk_wsplit(True))
    hunt(f"wsplit res L{lvl}", mk_wsplit(False))

    def mk_w4():
        x, r, y = act(g, c), act(g, c), act(g, c, False)
        return lambda: (K.conv3x3_winograd4(g, x, w4, c, c, y, res1=r), y)[1]

    hunt(f"winograd4 res (no split) L{lvl}", mk_w4)
    if lvl < 3:
        c2 = 2 * c
        g2 = K.geom(B, side // 2, side // 2)
        wd = (torch.randn(c2, c, 2, 2, generator=gen) / (2.0 * c ** 0.5)).to(dev)
        wdp = K.pack_down_bf16s_weight(wd)
        wdp3 = K.pack_down_bf16x3_weight(wd)
        wu = (torch.randn(c2, c, 2, 2, generator=gen) / (2.0 * c2 ** 0.5)).to(dev)
        wup = K.pack_up_bf16s_weight(wu)
        wuf = K.pack_up_weight(wu)

        def mk_down(which):
            def make():
                x, y = act(g, c), act(g2, c2, False)
                if which == "s":
                    return lambda: (K.down2x2_bf16s(g, g2, x, wdp, c, c2, y), y)[1]
                return lambda: (K.down2x2_bf16x3(g, g2, x, wdp3, c, c2, y), y)[1]
            return make

        def mk_up(which):
            def make():
                x, x2, y = act(g2, c2), act(g2, c2), act(g, c, False)
                if which == "s":
                    return lambda: (K.up2x2_bf16s(g2, g, x, x2, wup, c2, c, y), y)[1]
                return lambda: (K.up2x2(g2, g, x, x2, wuf, c2, c, y), y)[1]
            return make

        hunt(f"down2x2_bf16s L{lvl}", mk_down("s"))
        hunt(f"down2x2_bf16x3 L{lvl}", mk_down("3"))
        hunt(f"up2x2_bf16s L{lvl + 1}->L{lvl}", mk_up("s"))
        hunt(f"up2x2 fp32 L{lvl + 1}->L{lvl}", mk_up("f"))

# the whole model, both precisions: repeated lanes runs against each other
for prec in ("bf16split", "fp32"):
    den = dinv.models.DRUNet(3, 3, pretrained=None).to(dev).eval()
    den.conv_precision = prec
    x = torch.rand(16, 3, 256, 256, generator=gen).to(dev)
    with torch.no_grad():
        den.batch_lanes = 1
        ref = den(x, 0.1)
        den.batch_lanes = 2
        outs = [den(x, 0.1) for _ in range(20)]
    torch.cuda.synchronize()
    diff = [float((o - ref).abs().max()) for o in outs]
    print(json.dumps({"model": prec, "distinct_lane_outputs": len({d for d in diff}), "max_abs_diff_vs_one_lane": max(diff), "min": min(diff)}), flush=True)
