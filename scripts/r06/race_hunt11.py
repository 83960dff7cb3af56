"""which TERMS of the tail convolution get lost beside a bf16-split launch: one-hot weights over every (input channel, tap), output
channel `co`; prints the (channel, tap) pairs with wrong outputs, the wrong outputs' row within the wave's row group and what they hold"""
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
B, side, c = 8, 256, 64
cout = int(sys.argv[1]) if len(sys.argv) > 1 else 2
co = int(sys.argv[2]) if len(sys.argv) > 2 else 0
RB = 4
g = K.geom(B, side, side)


def fill(a, t):
    a[:, g.sl:g.sl + g.np].view(-1, B, g.hp, g.wp, 8)[:, :, 1:side + 1, 1:side + 1] = t.view(B, -1, 8, side, side).permute(1, 0, 3, 4, 2)
    return a


code = (torch.arange(c).view(1, c, 1, 1) + c * (torch.arange(side).view(1, 1, 1, side) + side * torch.arange(side).view(1, 1, side, 1))
        ).float().expand(B, c, side, side).contiguous() + 1.0
xa = fill(K.alloc(g, c, dev), code.to(dev))
x2a = K.alloc(g, c, dev)
yt = K.alloc(g, cout, dev)
w = (torch.randn(c, c, 3, 3, generator=gen) / 24).to(dev)
wws = K.pack_wsplit_weight(w)
xb, rb, yb = (fill(K.alloc(g, c, dev), torch.randn(B, c, side, side, generator=gen).to(dev)) for _ in range(3))
per_term, rows, values, lanes = collections.Counter(), collections.Counter(), collections.Counter(), collections.Counter()
for j in range(c):
    for t in range(9):
        wt = torch.zeros(cout, c, 3, 3)
        wt[co, j, t // 3, t % 3] = 1.0
        wtp = K.pack_tail_weight(wt.to(dev))
        with torch.cuda.stream(sA):
            K.conv3x3_tail(g, xa, wtp, c, cout, yt, x2=x2a)
        torch.cuda.synchronize()
        ref = yt.clone()
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
            got = yt[0, g.sl:g.sl + g.np].view(B, g.hp, g.wp, 8)
            exp = ref[0, g.sl:g.sl + g.np].view(B, g.hp, g.wp, 8)
            idx = (got != exp).nonzero()
            if len(idx):
                per_term[f"ch{j} (block {j // 8}, i {j % 8}) tap {t}"] += len(idx)
                rows.update(((idx[:, 1] - 1) % RB).tolist())
                lanes.update(((idx[:, 2] - 1) % 52 + 1).tolist())
                values.update(["zero" if v == 0 else "other" for v in got[tuple(idx.T)].tolist()])
                values.update([f"out channel {k}" for k in idx[:, 3].tolist()])
print(json.dumps({"cout": cout, "one_hot_out_channel": co, "terms_with_wrong_outputs": dict(per_term), "row_in_group": dict(sorted(rows.items())),
                  "lanes": dict(sorted(lanes.items())), "values": dict(values)}))
