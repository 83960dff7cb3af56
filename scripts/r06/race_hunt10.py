"""what the wrong outputs of the tail convolution ARE (scripts/r06/race_hunt8.py: wrong beside a bf16-split launch, last lanes of a strip,
channel 0): the input encodes (channel, row, column) as an exact integer and the weight is one-hot (one input channel, one tap, output
channel 0), so every output element names the input element it was computed from"""
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
g = K.geom(B, side, side)


def fill(a, t):
    a[:, g.sl:g.sl + g.np].view(-1, B, g.hp, g.wp, 8)[:, :, 1:side + 1, 1:side + 1] = t.view(B, -1, 8, side, side).permute(1, 0, 3, 4, 2)
    return a


def act(ch, fill_it=True):
    a = K.alloc(g, ch, dev)
    return fill(a, torch.randn(B, ch, side, side, generator=gen).to(dev)) if fill_it else a


code = (torch.arange(c).view(1, c, 1, 1) + c * (torch.arange(side).view(1, 1, 1, side) + side * torch.arange(side).view(1, 1, side, 1))
        ).float().expand(B, c, side, side).contiguous()      # exact in fp32: 64 * 256 * 256 < 2^24
xa = fill(K.alloc(g, c, dev), code.to(dev))
x2a = K.alloc(g, c, dev)                                       # zeros: the skip-add path is taken, the sum is the code
yt = K.alloc(g, cout, dev)
w = (torch.randn(c, c, 3, 3, generator=gen) / 24).to(dev)
wws = K.pack_wsplit_weight(w)
xb, rb, yb = act(c), act(c), act(c, False)
cats = collections.Counter()
examples = []
for j in (0, 1, 7, 8, 37, 63):
    for t in range(9):
        wt = torch.zeros(cout, c, 3, 3)
        wt[0, j, t // 3, t % 3] = 1.0
        wtp = K.pack_tail_weight(wt.to(dev))
        with torch.cuda.stream(sA):
            K.conv3x3_tail(g, xa, wtp, c, cout, yt, x2=x2a)
        torch.cuda.synchronize()
        ref = yt.clone()
        for it in range(4):
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
            for b_, r_, c_, ch_ in idx[:: max(1, len(idx) // 50)].tolist():
                gv, ev = got[b_, r_, c_, ch_].item(), exp[b_, r_, c_, ch_].item()

                def dec(v):
                    if v != int(v) or v < 0:
                        return None
                    v = int(v)
                    return (v % c, v // c // side, v // c % side)      # (channel, row, column) 0-based image coordinates
                de, dg = dec(ev), dec(gv)
                if gv == 0:
                    cat = "zero"
                elif dg is None:
                    cat = "not a code"
                elif de is None:
                    cat = "expected not a code?"
                else:
                    cat = f"dch={dg[0]-de[0]} drow={dg[1]-de[1]} dcol={dg[2]-de[2]}"
                cats[(ch_, cat)] += 1
                if len(examples) < 40:
                    examples.append({"j": j, "t": t, "at": (b_, r_ - 1, c_ - 1, ch_), "lane": (c_ - 1) % 52 + 1, "expected": ev, "got": gv, "exp_code": de, "got_code": dg})
print(json.dumps({"cout": cout, "categories": {f"ch{k[0]} {k[1]}": v for k, v in cats.most_common(40)}}))
for e in examples:
    print(json.dumps(e))
