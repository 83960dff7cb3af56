"""ms per launch of the tail convolution (64 -> 2 channels, skip add) at 320 x 320 for several batches"""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from deepinv_amd.hip import drunet as K
dev = torch.device("cuda:0")
for B in (32, 16, 4, 2):
    g = K.geom(B, 320, 320)
    x, x2, y = K.alloc(g, 64, dev).normal_(), K.alloc(g, 64, dev).normal_(), K.alloc(g, 2, dev)
    w = K.pack_tail_weight(torch.randn(2, 64, 3, 3, device=dev) / 24)
    for _ in range(5):
        K.conv3x3_tail(g, x, w, 64, 2, y, x2=x2)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(50):
        K.conv3x3_tail(g, x, w, 64, 2, y, x2=x2)
    e1.record(); torch.cuda.synchronize()
    print(json.dumps({"tag": sys.argv[1] if len(sys.argv) > 1 else "", "batch": B, "tail_ms": round(e0.elapsed_time(e1) / 50, 4)}), flush=True)
