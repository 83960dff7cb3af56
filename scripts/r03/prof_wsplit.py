"""one conv1 -> conv2 pair of csrc/drunet_wsplit.hip at DRUNet level argv[1], batch argv[2] (for rocprofv3 --pmc)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from deepinv_amd.hip import drunet as K
lvl, B = int(sys.argv[1]), int(sys.argv[2])
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 3
dev = torch.device("cuda:0")
c, H = 64 << lvl, 320 >> lvl
g = K.geom(B, H, H)
x, y, r, t = K.alloc(g, c, dev), K.alloc(g, c, dev), K.alloc(g, c, dev), K.alloc(g, c, dev)
x[:, g.sl:g.sl + g.np].view(-1, B, g.hp, g.wp, 8)[:, :, 1:H + 1, 1:H + 1].normal_()
r[:, g.sl:g.sl + g.np].view(-1, B, g.hp, g.wp, 8)[:, :, 1:H + 1, 1:H + 1].normal_()
ww = K.pack_wsplit_weight(torch.randn(c, c, 3, 3, device=dev) / (3 * c ** 0.5))
for _ in range(iters):
    K.conv3x3_wsplit(g, x, ww, c, c, t, relu=True)
    K.conv3x3_wsplit(g, t, ww, c, c, y, res1=r)
torch.cuda.synchronize()
print("ok")
