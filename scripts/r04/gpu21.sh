#!/bin/bash
# twenty-first hardware run: Radon adjoint with out-of-grid candidates moved away instead of masked, fan-beam adjoint with the tighter
# candidate window: tests and operator timings
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_tomography_gpu.py -q -m gpu 2>&1 | tail -2
timeout 300 python scripts/bench_ops.py radon 2>&1 | tail -5 | cut -c1-150
timeout 300 python - <<'P'
import torch, time, sys
sys.path.insert(0, '.')
import deepinv_amd as dinv
dev = torch.device('cuda:0')
B, W, A = 8, 512, 720
p = dinv.physics.Tomography(angles=A, img_width=W, circle=False, normalize=False, fan_beam=True, device=dev)
x = torch.rand(B, 1, W, W, device=dev)
y = p.A(x)
for name, fn in (("fan A", lambda: p.A(x)), ("fan A_adjoint", lambda: p.A_adjoint(y))):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): fn()
    torch.cuda.synchronize(); print(name, round((time.perf_counter() - t0) / 5 * 1e3, 3), "ms")
P
