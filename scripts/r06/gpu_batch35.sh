#!/bin/bash
# the strided LDS-tiled convolutions (Downsampling.A / A_adjoint / prox_l2): parity tests and timings
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -q -m gpu -k "blur or conv or downsampl or cfg5 or diffpir" 2>&1 | tail -3
timeout 600 python scripts/r06/bench_blur.py 2>&1 | grep "Downsampling" | cut -c1-200
timeout 300 python - <<'P'
import torch, sys
sys.path.insert(0, '.')
import deepinv_amd as dinv
dev = torch.device("cuda:0")
p = dinv.physics.Downsampling(img_size=(3, 256, 256), filter="bicubic", factor=4, padding="circular", device=dev)
x = torch.rand(16, 3, 256, 256, device=dev); y = p.A(x)
def t(fn, n=100):
    for _ in range(10): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return round(e0.elapsed_time(e1) / n, 4)
print({"A_ms": t(lambda: p.A(x)), "AT_ms": t(lambda: p.A_adjoint(y)), "prox_ms": t(lambda: p.prox_l2(x, y, 1.3))})
P
