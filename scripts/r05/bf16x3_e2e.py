"""DRUNet(2->2) forward at 320x320, conv_precision = "fp32", with the F(4x4) launches in their fp32-MFMA form and in their bf16 x 3
form (hip/drunet.py: FP32_WINOGRAD4_BF16X3), SUSTAINED (each point: >= 2 s of back-to-back forwards after 1 s of warm-up, so that the
package power limit has settled): ms per forward per batch.   python scripts/r05/bf16x3_e2e.py [batches...]"""
import json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import deepinv_amd as dinv
from deepinv_amd.hip import drunet as K

dev = torch.device("cuda:0")
den = dinv.models.DRUNet(2, 2, pretrained=None).to(dev).eval()
den.conv_precision = "fp32"
for B in [int(a) for a in sys.argv[1:]] or [4, 8, 16, 32]:
    x = torch.randn(B, 2, 320, 320, device=dev)
    row = {"batch": B}
    outs = {}
    for form in (False, True, False, True):
        K.FP32_WINOGRAD4_BF16X3 = form
        with torch.no_grad():
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < 1.0:
                outs[form] = den(x, 0.05)
                torch.cuda.synchronize()
            n, t0 = 0, time.perf_counter()
            while time.perf_counter() - t0 < 2.0:
                for _ in range(5):
                    den(x, 0.05)
                torch.cuda.synchronize()
                n += 5
            dt = (time.perf_counter() - t0) / n
        row.setdefault("bf16x3_ms" if form else "fp32_mfma_ms", []).append(round(dt * 1e3, 3))
    row["rel_diff_of_outputs"] = float((outs[True] - outs[False]).norm() / outs[False].norm())
    print(json.dumps(row), flush=True)
