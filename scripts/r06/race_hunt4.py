"""conv3x3_wsplit at level 0 (64 channels, 8 x 256 x 256) on stream A while stream B runs ANOTHER kernel of the network: which partner makes
A's output differ from its solo result?"""
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
B, side, c = 8, 256, 64
g = K.geom(B, side, side)
g1 = K.geom(B, side // 2, side // 2)


def act(gg, ch, fill=True):
    a = K.alloc(gg, ch, dev)
    if fill:
        H, W = gg.height, gg.width
        t = torch.randn(B, ch, H, W, generator=gen).relu_().to(dev)
        a[:, gg.sl:gg.sl + gg.np].view(-1, B, gg.hp, gg.wp, 8)[:, :, 1:H + 1, 1:W + 1] = t.view(B, -1, 8, H, W).permute(1, 0, 3, 4, 2)
    return a


w = (torch.randn(c, c, 3, 3, generator=gen) / (3.0 * c ** 0.5)).to(dev)
wws = K.pack_wsplit_weight(w)
xa, ra, ya = act(g, c), act(g, c), act(g, c, False)
ref = {}
for mode in ("relu", "res"):
    with torch.cuda.stream(sA):
        K.conv3x3_wsplit(g, xa, wws, c, c, ya, res1=ra if mode == "res" else None, relu=mode == "relu")
    torch.cuda.synchronize()
    ref[mode] = ya.clone()

# partners on stream B (own buffers)
xb, rb, yb = act(g, c), act(g, c), act(g, c, False)
wd, ci, co = K.pack_conv3x3_weight(w)
xin = act(g, 8)
wh = (torch.randn(c, 8, 3, 3, generator=gen) / 8).to(dev)
whp, hci, hco = K.pack_conv3x3_weight(wh)
wdn = (torch.randn(2 * c, c, 2, 2, generator=gen) / 16).to(dev)
wdn_f, wdn_s, wdn_3 = K.pack_down_weight(wdn), K.pack_down_bf16s_weight(wdn), K.pack_down_bf16x3_weight(wdn)
y1 = act(g1, 2 * c, False)
x1, x1b = act(g1, 2 * c), act(g1, 2 * c)
wup = (torch.randn(2 * c, c, 2, 2, generator=gen) / 16).to(dev)
wup_f, wup_s = K.pack_up_weight(wup), K.pack_up_bf16s_weight(wup)
wt = (torch.randn(3, c, 3, 3, generator=gen) / 24).to(dev)
wtp = K.pack_tail_weight(wt)
yt = K.alloc(g, 3, dev)
ximg = torch.rand(B, 3, side, side, generator=gen).to(dev)
yimg = torch.empty(B, 3, side, side, device=dev)
w4 = K.pack_winograd4_weight(w)
partners = {
    "wsplit (same kernel)": lambda: K.conv3x3_wsplit(g, xb, wws, c, c, yb, relu=True),
    "head conv3x3 direct": lambda: K.conv3x3(g, xin, whp, hci, hco, yb, cin_valid=4),
    "conv3x3 direct 64->64": lambda: K.conv3x3(g, xb, wd, ci, co, yb, relu=True),
    "winograd4": lambda: K.conv3x3_winograd4(g, xb, w4, c, c, yb, relu=True),
    "down2x2 fp32": lambda: K.down2x2(g, g1, xb, wdn_f, c, 2 * c, y1),
    "down2x2_bf16s": lambda: K.down2x2_bf16s(g, g1, xb, wdn_s, c, 2 * c, y1),
    "up2x2 fp32": lambda: K.up2x2(g1, g, x1, x1b, wup_f, 2 * c, c, yb),
    "up2x2_bf16s": lambda: K.up2x2_bf16s(g1, g, x1, x1b, wup_s, 2 * c, c, yb),
    "tail": lambda: K.conv3x3_tail(g, xb, wtp, c, 3, yt, x2=rb),
    "pack": lambda: K.pack_input(g, ximg, 0.1, xin),
    "unpack": lambda: K.unpack_output(g, yt, 3, yimg),
}
for pname, partner in partners.items():
    for mode in ("relu", "res"):
        bad = 0
        for it in range(40):
            with torch.cuda.stream(sB):
                for _ in range(3):
                    partner()
            with torch.cuda.stream(sA):
                K.conv3x3_wsplit(g, xa, wws, c, c, ya, res1=ra if mode == "res" else None, relu=mode == "relu")
            with torch.cuda.stream(sB):
                for _ in range(3):
                    partner()
            torch.cuda.synchronize()
            bad += 0 if torch.equal(ya, ref[mode]) else 1
        print(json.dumps({"partner_on_other_stream": pname, "wsplit_mode": mode, "mismatches_of_40": bad}), flush=True)
