"""time the bf16-split conv (csrc/drunet_bf16s.hip) at DRUNet level argv[1] for batch argv[2]; used under rocprofv3 --pmc"""
import os, sys, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from deepinv_amd.hip import drunet as K

lvl, B = int(sys.argv[1]), int(sys.argv[2])
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 20
dev = torch.device("cuda:0")
c, H = 64 << lvl, 320 >> lvl
g = K.geom(B, H, H)
x, y = K.alloc(g, c, dev), K.alloc(g, c, dev)
x.normal_()
w = torch.randn(c, c, 3, 3, device=dev) / (3 * c ** 0.5)
ws = K.pack_bf16s_weight(w)
for _ in range(3):
    K.conv3x3_bf16s(g, x, ws, c, c, y, relu=True)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    K.conv3x3_bf16s(g, x, ws, c, c, y, relu=True)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / iters
fl = 2.0 * 9 * c * c * B * H * H
print(json.dumps({"lvl": lvl, "B": B, "ms": ms, "effTF": fl / ms / 1e9, "executed_TF": 3 * fl / ms / 1e9,
                  "frac_bf16_peak": 3 * fl / ms / 1e9 / 2500}))
