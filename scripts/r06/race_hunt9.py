"""details of the tail kernel's differing pixels when a wsplit kernel shares the device: rows mod RB, values, and which single input
tap (channel block, row offset, column offset) would explain got - ref if it were dropped or doubled"""
import collections
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
B, side, c, cout = 8, 256, 64, 3
g = K.geom(B, side, side)
xt = torch.randn(B, c, side, side, generator=gen)
x2t = torch.randn(B, c, side, side, generator=gen)


def act(t):
    a = K.alloc(g, t.shape[1], dev)
    a[:, g.sl:g.sl + g.np].view(-1, B, g.hp, g.wp, 8)[:, :, 1:side + 1, 1:side + 1] = t.to(dev).view(B, -1, 8, side, side).permute(1, 0, 3, 4, 2)
    return a


xa, x2a = act(xt), act(x2t)
wt = (torch.randn(cout, c, 3, 3, generator=gen) / 24)
wtp = K.pack_tail_weight(wt.to(dev))
yt = K.alloc(g, cout, dev)
w = (torch.randn(c, c, 3, 3, generator=gen) / 24).to(dev)
wws = K.pack_wsplit_weight(w)
xb, rb, yb = act(torch.randn(B, c, side, side, generator=gen)), act(torch.randn(B, c, side, side, generator=gen)), K.alloc(g, c, dev)
with torch.cuda.stream(sA):
    K.conv3x3_tail(g, xa, wtp, c, cout, yt, x2=x2a)
torch.cuda.synchronize()
ref = yt.clone()
exact = torch.nn.functional.conv2d((xt + x2t).double(), wt.double(), padding=1)      # [B, 3, H, W]
for it in range(3):
    with torch.cuda.stream(sB):
        for _ in range(2):
            K.conv3x3_wsplit(g, xb, wws, c, c, yb, res1=rb)
    with torch.cuda.stream(sA):
        K.conv3x3_tail(g, xa, wtp, c, cout, yt, x2=x2a)
    with torch.cuda.stream(sB):
        for _ in range(2):
            K.conv3x3_wsplit(g, xb, wws, c, c, yb, res1=rb)
    torch.cuda.synchronize()
    view = lambda t: t[0, g.sl:g.sl + g.np].view(B, g.hp, g.wp, 8)[:, 1:side + 1, 1:side + 1, :cout].cpu()
    got, rf = view(yt), view(ref)
    d = (got - rf)
    idx = (d != 0).nonzero()
    rows = collections.Counter((idx[:, 1] % 8).tolist())
    print(json.dumps({"run": it, "n": int(idx.shape[0]), "rows_mod_8": dict(sorted(rows.items())), "units": dict(collections.Counter(idx[:, 0].tolist())),
                      "ref_err_vs_fp64": float((rf.permute(0, 3, 1, 2).double() - exact).abs().max()),
                      "got_err_vs_fp64": float((got.permute(0, 3, 1, 2).double() - exact).abs().max())}), flush=True)
    S = (xt + x2t)
    for (b, r, cc, co) in idx[:6].tolist():
        delta = float(d[b, r, cc, co])
        # which single term w[co, ci, dy, dx] * S[b, ci, r + dy - 1, cc + dx - 1] equals -delta (dropped) or +delta (doubled)?
        best = None
        for dy in range(3):
            for dx in range(3):
                rr, c2 = r + dy - 1, cc + dx - 1
                if 0 <= rr < side and 0 <= c2 < side:
                    terms = wt[co, :, dy, dx] * S[b, :, rr, c2]
                    # sums over one 8-channel block
                    blocks = terms.view(8, 8).sum(1)
                    for cb in range(8):
                        for sign in (1, -1):
                            e = abs(float(blocks[cb]) * sign - delta)
                            if best is None or e < best[0]:
                                best = (e, cb, dy, dx, sign, float(blocks[cb]))
        print(json.dumps({"pixel": [b, r, cc, co], "ref": float(rf[b, r, cc, co]), "got": float(got[b, r, cc, co]), "delta": delta,
                          "closest_block_term": {"abs_residual": best[0], "channel_block": best[1], "dy": best[2], "dx": best[3], "sign": best[4], "value": best[5]}}), flush=True)
