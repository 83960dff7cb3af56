"""phase timing of conv3x3_wino_kernel (needs a -DDINV_WINO_TIMING build): argv lvl B [channels]"""
import os, sys, ctypes
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from deepinv_amd.hip import drunet as K, lib

lvl, B = int(sys.argv[1]), int(sys.argv[2])
c, H = 64 << lvl, 320 >> lvl
if len(sys.argv) > 3:
    c = int(sys.argv[3])
res = len(sys.argv) > 4
dev = torch.device("cuda:0")
g = K.geom(B, H, H)
x, y, r = K.alloc(g, c, dev), K.alloc(g, c, dev), K.alloc(g, c, dev)
x.normal_()
ww = K.pack_winograd_weight(torch.randn(c, c, 3, 3, device=dev) / (3 * c ** 0.5))
dbg = torch.zeros(256 * 4 * 32, dtype=torch.int64, device=dev)
l = lib()
l.dinv_debug_wino_timing.argtypes = [ctypes.c_void_p]
for _ in range(3):
    K.conv3x3_winograd(g, x, ww, c, c, y, relu=not res, res1=r if res else None)
l.dinv_debug_wino_timing(ctypes.c_void_p(dbg.data_ptr()))
K.conv3x3_winograd(g, x, ww, c, c, y, relu=not res, res1=r if res else None)
torch.cuda.synchronize()
l.dinv_debug_wino_timing(None)
d32 = dbg.view(256, 4, 32).cpu().double()
d = d32[:, :, :8]
names = ["prologue", "main loop", "offsets+res+next-tile loads", "barrier", "s-calc+exch write+barrier", "exch read+barrier", "finalise+stores"]
ok = d[:, :, 7] > 0
for k in range(4):
    m = ok[:, k]
    if m.sum() == 0:
        continue
    seg = (d[m, k, 1:] - d[m, k, :-1])
    print(f"tile {k}: n={int(m.sum())} total={float((d[m,k,7]-d[m,k,0]).mean()):.0f} | " +
          " | ".join(f"{n} {float(seg[:, i].mean()):.0f}" for i, n in enumerate(names)))
    if k > 0:
        gap = d[m, k, 0] - d[m, k - 1, 7]
        print(f"        gap from previous tile end: {float(gap.mean()):.0f}")

m = ok[:, 1]
b = d32[m, 1, 8:24].view(-1, 2, 8)
print("block 2/3 of tile 1 (cycles): step0 %.0f | step1+stash %.0f | barrier %.0f | step2 %.0f | step3 (to next block start) %.0f" % (
    float((b[:, 0, 1] - b[:, 0, 0]).mean()), float((b[:, 0, 2] - b[:, 0, 1]).mean()), float((b[:, 0, 3] - b[:, 0, 2]).mean()),
    float((b[:, 0, 4] - b[:, 0, 3]).mean()), float((b[:, 1, 0] - b[:, 0, 4]).mean())))
