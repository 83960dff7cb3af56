#!/bin/bash
# twenty-seventh hardware run: the bf16-split transposed convolution without its 80 bytes of scratch per lane (waves_per_eu(2, 2)):
# the tests that run it, the DRUNet forward of the bf16-split setting
cd $GRAFT_REPO_ROOT
timeout 200 python -m pytest tests/test_drunet_gpu.py -q -m gpu -k "drunet_matches_oracle or default_precision or drunet3d" 2>&1 | tail -2
timeout 100 python - <<'P'
import torch, time, sys
sys.path.insert(0, '.')
import deepinv_amd as dinv
dev = torch.device('cuda:0')
m = dinv.models.DRUNet(2, 2, pretrained=None).to(dev).eval()
x = torch.rand(32, 2, 320, 320, device=dev)
for prec in ("bf16split", "fp32"):
    m.conv_precision = prec
    with torch.no_grad():
        for _ in range(3): m(x, 0.05)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): m(x, 0.05)
        torch.cuda.synchronize(); print(prec, "DRUNet forward ms", round((time.perf_counter() - t0) * 100, 3))
P
