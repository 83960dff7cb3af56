"""a chain of four ResBlocks at level 0 (conv1 relu -> conv2 + residual, ping-pong buffers, as DRUNet._res_chain) on stream A while stream B
runs the same chain on its own buffers: does A's result differ from its solo result?  Variants: kernel family (wsplit / winograd4), and a
device-wide synchronisation point between the two convolutions of a block (events do not change the hardware ordering of one stream)."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from deepinv_amd.hip import drunet as K  # noqa: E402

dev = torch.device("cuda:0")
sA, sB = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
gen = torch.Generator().manual_seed(0)
B, side, c = 8, int(sys.argv[1]) if len(sys.argv) > 1 else 256, int(sys.argv[2]) if len(sys.argv) > 2 else 64
g = K.geom(B, side, side)


def act(fill=True):
    a = K.alloc(g, c, dev)
    if fill:
        t = torch.randn(B, c, side, side, generator=gen).relu_().to(dev)
        a[:, g.sl:g.sl + g.np].view(-1, B, g.hp, g.wp, 8)[:, :, 1:side + 1, 1:side + 1] = t.view(B, -1, 8, side, side).permute(1, 0, 3, 4, 2)
    return a


ws = [(torch.randn(c, c, 3, 3, generator=gen) / (3.0 * c ** 0.5)).to(dev) for _ in range(8)]
packs = {"wsplit": [K.pack_wsplit_weight(w) for w in ws], "winograd4": [K.pack_winograd4_weight(w) for w in ws]}


def chain(kind, bufs):
    x, a, b, t = bufs
    cur, pp = x, [a, b]
    for i in range(4):
        dst = pp[i % 2]
        if kind == "wsplit":
            K.conv3x3_wsplit(g, cur, packs[kind][2 * i], c, c, t, relu=True)
            K.conv3x3_wsplit(g, t, packs[kind][2 * i + 1], c, c, dst, res1=cur)
        else:
            K.conv3x3_winograd4(g, cur, packs[kind][2 * i], c, c, t, relu=True)
            K.conv3x3_winograd4(g, t, packs[kind][2 * i + 1], c, c, dst, res1=cur)
        cur = dst
    return cur


for kind in ("wsplit", "winograd4"):
    bufA = [act(), act(False), act(False), act(False)]
    bufB = [act(), act(False), act(False), act(False)]
    with torch.cuda.stream(sA):
        refA = chain(kind, bufA).clone()
    torch.cuda.synchronize()
    with torch.cuda.stream(sB):
        refB = chain(kind, bufB).clone()
    torch.cuda.synchronize()
    for mode in ("A alone, repeated", "A and B concurrently"):
        bad, worst = 0, 0.0
        for it in range(40):
            with torch.cuda.stream(sA):
                oa = chain(kind, bufA)
            if mode.startswith("A and B"):
                with torch.cuda.stream(sB):
                    ob = chain(kind, bufB)
            torch.cuda.synchronize()
            if not torch.equal(oa, refA):
                bad += 1
                worst = max(worst, float((oa - refA).abs().max()))
            if mode.startswith("A and B") and not torch.equal(ob, refB):
                bad += 1
                worst = max(worst, float((ob - refB).abs().max()))
        print(json.dumps({"kernels": kind, "side": side, "channels": c, "mode": mode, "mismatching_results": bad, "max_abs_diff": worst}), flush=True)
