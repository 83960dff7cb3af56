"""Fit t = tiles_per_CU * (F + nsub * s) for the bf16-split conv: same 320x320 B=32 pixels, different (cin, cout)."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepinv_amd.hip import drunet as K  # noqa: E402
from bench_ops import timeit  # noqa: E402

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
H = int(sys.argv[2]) if len(sys.argv) > 2 else 320
g = K.geom(B, H, H)
for cin, cout in ((16, 64), (32, 64), (64, 64), (128, 64), (256, 64), (64, 128), (128, 128)):
    x, y = K.alloc(g, cin, dev), K.alloc(g, cout, dev)
    x.normal_()
    w = torch.randn(cout, cin, 3, 3, device=dev) / (3 * cin ** 0.5)
    ws = K.pack_bf16s_weight(w)
    t = timeit(lambda: K.conv3x3_bf16s(g, x, ws, cin, cout, y, relu=True), iters=20, warmup=3)
    tp = 256 if os.environ.get("DINV_BF16S_WAVES") == "4" else 512
    wgs = -(-g.np // tp) * (cout // 64)
    print(json.dumps({"cin": cin, "cout": cout, "ms": round(t * 1e3, 4), "nsub": 3 * cin // 16, "wgs": wgs,
                      "us_per_wg_slot": round(t * 1e6 / (wgs / (256 * (2 if tp == 256 else 1))), 2)}))
